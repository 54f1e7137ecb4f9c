"""Diagnostic (GPU box): run-to-run spread of the network-level gradient error statistics that
tests/test_gpu_network.py::test_segnet_backward_matches_oracle asserts on (split-K atomics make the HIP path's
rounding realisation, hence its ReLU / max-pool flips, vary from run to run)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import seeded_inputs, seeded_state, set_dropout  # noqa: E402
from oracle import heatnet_ref as R  # noqa: E402


def main():
    from heatnet_amd import train_step
    from heatnet_amd.pspnet import PSPNet
    dev = torch.device("cuda:0")
    rgb, ir, lab = seeded_inputs(2, 64, 96, seed=3)
    print("seed run   median-ratio  p90-ratio  max-ratio   (hip error / cpu-fp32 error, both vs fp64)")
    for seed in (1, 2, 3):
        grads = {}
        for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
            ref = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
            state = seeded_state(ref, seed=seed)
            ref = ref.to(dt)
            ref.load_state_dict({k: (v.to(dt) if v.is_floating_point() else v) for k, v in state.items()})
            set_dropout(ref, 0.0)
            ref.train()
            torch.nn.functional.cross_entropy(ref(rgb.to(dt), ir.to(dt))[0], lab).backward()
            grads[name] = {k: p.grad.double() for k, p in ref.named_parameters()}
        floor = 1e-6 * max(float(g.norm()) for g in grads["f64"].values())
        keys = [k for k, t in grads["f64"].items() if float(t.norm()) >= floor]
        cpu = np.array([float((grads["f32"][k] - grads["f64"][k]).norm() / grads["f64"][k].norm()) for k in keys])
        for run in range(5):
            net = PSPNet(backend="resnet50", pretrained=False, late_fusion=True, in_channels=4)
            net.load_state_dict(state)
            set_dropout(net, 0.0)
            net = net.to(dev).train()
            train_step.segmentation_loss(net(rgb.to(dev), ir.to(dev))[0], lab.to(dev)).backward()
            got = dict(net.named_parameters())
            hip = np.array([float((got[k].grad.detach().cpu().double() - grads["f64"][k]).norm() / grads["f64"][k].norm())
                            for k in keys])
            print("%4d %3d   %10.2f %10.2f %10.2f" % (seed, run, np.median(hip) / np.median(cpu),
                                                    np.percentile(hip, 90) / np.percentile(cpu, 90), hip.max() / cpu.max()))


if __name__ == "__main__":
    main()
