"""Diagnostic (GPU box): parameter-gradient error of the HIP path and of the fp32 CPU oracle against an fp64 oracle."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import seeded_inputs, seeded_state, set_dropout  # noqa: E402
from oracle import heatnet_ref as R  # noqa: E402


def main():
    n, h, w, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    from heatnet_amd.pspnet import PSPNet
    from heatnet_amd import train_step
    dev = torch.device("cuda:0")
    rgb, ir, lab = seeded_inputs(n, h, w, seed=3)
    nets = {}
    for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
        net = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
        state = seeded_state(net, seed=seed)
        net = net.to(dt)
        net.load_state_dict({k: (v.to(dt) if v.is_floating_point() else v) for k, v in state.items()})
        set_dropout(net, 0.0)
        net.train()
        loss = torch.nn.functional.cross_entropy(net(rgb.to(dt), ir.to(dt))[0], lab)
        loss.backward()
        nets[name] = {k: p.grad.double() for k, p in net.named_parameters()}
    net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
    net.load_state_dict(state)
    set_dropout(net, 0.0)
    net = net.to(dev).train()
    loss = train_step.segmentation_loss(net(rgb.to(dev), ir.to(dev))[0], lab.to(dev))
    loss.backward()
    hip = {k: p.grad.detach().cpu().double() for k, p in net.named_parameters()}
    rows = []
    for k in hip:
        t = nets["f64"][k]
        rows.append((float((hip[k] - t).norm() / (t.norm() + 1e-30)), float((nets["f32"][k] - t).norm() / (t.norm() + 1e-30)),
                     float(t.norm()), k))
    rows.sort(reverse=True)
    print("%-44s %10s %12s %12s" % ("param", "|g|", "hip-f64 rel", "cpu32-f64 rel"))
    for r in rows[:25]:
        print("%-44s %10.3e %12.3e %12.3e" % (r[3], r[2], r[0], r[1]))
    import statistics
    print("median rel err: hip %.3e cpu32 %.3e" % (statistics.median(r[0] for r in rows), statistics.median(r[1] for r in rows)))


if __name__ == "__main__":
    main()
