"""Diagnostic (GPU box): the same 30 RMSprop iterations (fixed mini-batch, Dropout off) on the CPU oracle, on the HIP path
under the fp32-MFMA arithmetic, under bf16x3 and under the mixed-precision bf16 arithmetic (configs[4]) -- do the loss
curves track each other?

usage: python tools/train_curve.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import seeded_inputs, seeded_state, set_dropout  # noqa: E402
from oracle import heatnet_ref as R  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    from heatnet_amd import ops, optim, train_step
    from heatnet_amd.pspnet import PSPNet
    dev = torch.device("cuda:0")
    rgb, ir, lab = seeded_inputs(2, 128, 256, seed=3)
    ref = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    state = seeded_state(ref, seed=1)
    curves = {}
    ref.load_state_dict(state)
    set_dropout(ref, 0.0)
    ref.train()
    opt = torch.optim.RMSprop(ref.parameters(), lr=1e-4)
    curves["cpu fp32 oracle"] = []
    for _ in range(steps):
        opt.zero_grad()
        loss = torch.nn.functional.cross_entropy(ref(rgb, ir)[0], lab)
        loss.backward()
        opt.step()
        curves["cpu fp32 oracle"].append(float(loss.detach()))
    for math in ("f32", "bf16x3", "f16x2", "bf16"):
        ops.set_conv_math(math)
        net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
        net.load_state_dict(state)
        set_dropout(net, 0.0)
        net = net.to(dev).train()
        opt = optim.FlatRMSprop(optim.arenas_for(net), lr=1e-4)
        d = [t.to(dev) for t in (rgb, ir, lab)]
        curves["hip " + math] = [float(train_step.segmentation_step(net, opt, d[0], d[1], d[2])) for _ in range(steps)]
    names = list(curves)
    print("step  " + "  ".join("%-16s" % n for n in names) + "  |hip - cpu|: f32, bf16x3, f16x2, bf16")
    for i in range(steps):
        v = [curves[n][i] for n in names]
        print("%4d  " % i + "  ".join("%-16.6f" % x for x in v) + "  " + "  ".join("%.2e" % abs(x - v[0]) for x in v[1:]))


if __name__ == "__main__":
    main()
