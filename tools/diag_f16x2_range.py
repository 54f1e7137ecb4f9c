"""f16x2 head-room on the network's own tensors (GPU box): for every convolution operand of one training step -- activations in
the forward pass, gradients in the backward pass -- log2(max|t| / rms t) and the fraction of non-zero elements that lie more
than 2^-18 below the tensor's maximum (those keep fewer than 22 significant bits under f16x2).
usage: python tools/diag_f16x2_range.py [n h w]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import seeded_inputs, seeded_state, set_dropout  # noqa: E402
from heatnet_amd import ops, optim, train_step  # noqa: E402
from heatnet_amd.pspnet import PSPNet  # noqa: E402
from oracle import heatnet_ref as R  # noqa: E402


def main():
    n, h, w = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (2, 320, 704)
    dev = torch.device("cuda:0")
    ops.set_conv_math("f16x2")
    rgb, ir, lab = seeded_inputs(n, h, w, seed=5)
    ref = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    state = seeded_state(ref, seed=1)
    net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
    net.load_state_dict(state)
    set_dropout(net, 0.0)
    net = net.to(dev).train()
    opt = optim.FlatRMSprop(optim.arenas_for(net), lr=1e-4)
    rows = []
    orig = ops.tensor_scale

    def spy(t, ld=None, owner=None):
        with torch.no_grad():
            a = t.detach().abs().float()
            mx = float(a.max())
            nz = a[a > 0]
            rms = float(a.pow(2).mean().sqrt())
            low = float((nz < mx * 2.0 ** -18).float().mean()) if nz.numel() else 0.0
            rows.append(("", tuple(t.shape), mx, rms, low))
        return orig(t, ld, owner)

    ops.tensor_scale = spy
    d = [t.to(dev) for t in (rgb, ir, lab)]
    for _ in range(2):
        rows.clear()
        train_step.segmentation_step(net, opt, d[0], d[1], d[2])
    ops.tensor_scale = orig
    import math
    worst = sorted(rows, key=lambda r: -(math.log2(r[2] / r[3]) if r[3] > 0 else 0))
    print("conv operands seen in one training step at %dx%dx%d: %d" % (n, h, w, len(rows)))
    print("%-26s %12s %12s %14s %22s" % ("shape", "max", "rms", "log2(max/rms)", "frac below max*2^-18"))
    for r in worst[:12]:
        print("%-26s %12.3e %12.3e %14.1f %22.2e" % (str(r[1]), r[2], r[3], math.log2(r[2] / r[3]) if r[3] > 0 else 0.0, r[4]))
    print("largest fraction of non-zero elements below max * 2^-18 in any operand: %.3e" % max(r[4] for r in rows))
    mags = sorted(r[2] for r in rows)
    print("operand maxima span %.2e ... %.2e" % (mags[0], mags[-1]))


if __name__ == "__main__":
    main()
