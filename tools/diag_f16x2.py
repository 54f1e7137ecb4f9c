"""f16x2 accuracy diagnostics (GPU box): rms error against an fp64 convolution for the three conv arithmetics, over reduction
lengths, gradient magnitudes and heavy-tailed inputs.  usage: HEATNET_F16X2_MIN_K=0 python tools/diag_f16x2.py"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HEATNET_F16X2_MIN_K", "0")
from heatnet_amd import ops  # noqa: E402


def rnd(*shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def main():
    dev = torch.device("cuda:0")
    cases = [  # n, cin, h, w, cout, k, stride, pad, dil, gy scale, tail
        (2, 64, 40, 48, 256, 1, 1, 0, 1, 1.0, 0), (2, 128, 40, 48, 512, 1, 1, 0, 1, 1.0, 0), (2, 256, 40, 48, 64, 1, 1, 0, 1, 1.0, 0),
        (2, 64, 40, 48, 64, 3, 1, 1, 1, 1.0, 0), (2, 512, 24, 40, 128, 1, 1, 0, 1, 1.0, 0), (2, 256, 24, 40, 256, 3, 1, 2, 2, 1.0, 0),
        (2, 1024, 20, 24, 256, 1, 1, 0, 1, 1.0, 0), (2, 128, 32, 48, 128, 3, 2, 1, 1, 1.0, 0), (1, 2048, 20, 24, 512, 1, 1, 0, 1, 1.0, 0),
        (2, 256, 24, 40, 256, 3, 1, 2, 2, 1e-6, 0), (2, 256, 24, 40, 256, 3, 1, 2, 2, 1e4, 0),
        (2, 256, 24, 40, 256, 3, 1, 2, 2, 1.0, 1), (2, 256, 24, 40, 256, 3, 1, 2, 2, 1.0, 2)]
    print("%-44s %-7s %10s %10s %10s   ratio to f32 (y dx dw)" % ("case", "math", "y", "dx", "dw"))
    for case in cases:
        n, cin, h, w, cout, k, stride, pad, dil, gscale, tail = case
        x = F.relu(rnd(n, cin, h, w, seed=31))
        if tail == 1:  # a few huge outliers: max / typical = 2^12
            x.view(-1)[::9973] *= 4096.0
        if tail == 2:  # one absurd outlier: max / typical = 2^22
            x.view(-1)[12345] = 4.0e6
        wt = rnd(cout, cin, k, k, seed=32, scale=1.0 / np.sqrt(cin * k * k))
        xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
        yr = F.conv2d(xr, wr, None, stride=stride, padding=pad, dilation=dil)
        gy = rnd(*yr.shape, seed=33) * gscale
        yr.backward(gy.double())
        errs = {}
        for math in ("f32", "bf16x3", "f16x2"):
            prev = ops.set_conv_math(math)
            try:
                xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
                wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
                y = ops.conv2d(xd, wd, None, stride, pad, dil)
                y.backward(gy.to(dev))
                errs[math] = tuple(float((a.detach().cpu().double() - b.detach()).pow(2).mean().sqrt() / b.detach().pow(2).mean().sqrt())
                                   for a, b in ((y, yr), (xd.grad, xr.grad), (wd.grad, wr.grad)))
            finally:
                ops.set_conv_math(prev)
        for math in ("f32", "bf16x3", "f16x2"):
            e = errs[math]
            r = tuple(a / b for a, b in zip(e, errs["f32"]))
            print("%-44s %-7s %10.3e %10.3e %10.3e   %.2f %.2f %.2f" % (str(case), math, e[0], e[1], e[2], r[0], r[1], r[2]))


if __name__ == "__main__":
    main()
