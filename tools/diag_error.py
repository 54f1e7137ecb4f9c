"""Diagnostic (GPU box): per-stage error of the HIP path and of the fp32 CPU oracle against an fp64 oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import seeded_inputs, seeded_state, set_dropout  # noqa: E402
from oracle import heatnet_ref as R  # noqa: E402


def stages(net, rgb, ir, names):
    outs = {}
    hooks = []
    for name in names:
        mod = dict(net.named_modules())[name]
        hooks.append(mod.register_forward_hook(lambda m, i, o, name=name: outs.__setitem__(name, o.detach())))
    logits, feats, _ = net(rgb, ir)
    for h in hooks:
        h.remove()
    outs["logits"] = logits.detach()
    return outs


def main():
    n, h, w, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from heatnet_amd.pspnet import PSPNet
    dev = torch.device("cuda:0")
    names = ["feats.maxpool", "feats.layer1", "feats.layer2", "feats.layer3.0", "feats.layer3", "feats.layer4.0",
             "feats.layer4", "psp", "up_1", "up_2", "up_3"]
    rgb, ir, _ = seeded_inputs(n, h, w, seed=5)
    ref32 = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    state = seeded_state(ref32, seed=seed)
    ref32.load_state_dict(state)
    set_dropout(ref32, 0.0)
    ref32.train()
    ref64 = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4).double()
    ref64.load_state_dict({k: (v.double() if v.is_floating_point() else v) for k, v in state.items()})
    set_dropout(ref64, 0.0)
    ref64.train()
    net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
    net.load_state_dict(state)
    set_dropout(net, 0.0)
    net = net.to(dev).train()
    rnames = [x for x in names if x != "feats.maxpool"]
    from heatnet_amd import ops
    maths = os.environ.get("DIAG_MATHS", ops.CONV_MATH).split(",")
    with torch.no_grad():
        o64 = stages(ref64, rgb.double(), ir.double(), rnames)
        o32 = stages(ref32, rgb, ir, rnames)
        ohip = {}
        for m in maths:
            ops.set_conv_math(m)
            ohip[m] = stages(net, rgb.to(dev), ir.to(dev), rnames)
    print("max |error| per stage against the fp64 oracle (DIAG_MATHS=%s)" % ",".join(maths))
    print("%-16s %10s %12s " % ("stage", "rms", "cpu32-f64") + " ".join("%12s" % ("hip " + m) for m in maths))
    for k in rnames + ["logits"]:
        t = o64[k]
        a = o32[k].double()
        print("%-16s %10.3e %12.3e " % (k, float(t.pow(2).mean().sqrt()), float((a - t).abs().max())) +
              " ".join("%12.3e" % float((ohip[m][k].cpu().double() - t).abs().max()) for m in maths))


if __name__ == "__main__":
    main()
