"""Microbenchmark (GPU box): the f16x2 convolution kernels that split the fp32 operand in registers (hn_conv2d_*_scaled) against
the all-LDS-DMA kernels on plane tensors (hn_conv2d_*_planes), per layer shape of the network.

usage: python tools/bench_planes.py [shape ...]   shape = N,H,W,Cin,Cout,K,stride,dil   (pad = dil*(K-1)//2)
env:   KINDS=fwd,dgrad,wgrad   ZERO=1 (all-zero operands: no data-dependent power limit)
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heatnet_amd import _lib, ops  # noqa: E402

DEFAULT = ["8,40,88,512,512,3,1,4", "8,40,88,256,256,3,1,2", "8,80,176,1024,256,3,1,1", "8,40,88,512,2048,1,1,1",
           "8,40,88,2048,512,1,1,1", "8,40,88,1024,256,1,1,1", "8,40,88,256,1024,1,1,1", "8,80,176,64,64,3,1,1",
           "8,80,176,64,256,1,1,1", "8,80,176,256,64,1,1,1", "8,40,88,128,128,3,1,1", "8,40,88,128,512,1,1,1",
           "8,160,352,256,64,3,1,1", "8,320,704,64,64,3,1,1", "8,80,176,128,128,3,2,1"]


def timeit(fn, reps=5, rounds=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    best = float("inf")
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    shapes = [a for a in sys.argv[1:]] or DEFAULT
    kinds = os.environ.get("KINDS", "fwd,dgrad,wgrad").split(",")
    zero = os.environ.get("ZERO") == "1"
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    V = ops._variant
    old = {"two-phase": V(0, 0, 0, 0), "dmaB128": V(3, 0, 0, 0), "dmaB256": V(3, 0, 3, 0)}
    print("TFLOP/s of fp32 work; 'scaled' = fp32 tensors split in the kernel, 'planes' = plane tensors, all by LDS-DMA")
    for sh in shapes:
        n, h, w, cin, cout, k, stride, dil = [int(v) for v in sh.split(",")]
        pad = dil * (k - 1) // 2
        ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w, k, stride, pad, dil)
        x = torch.relu(torch.randn(n, h, w, cin, device=dev))
        wt = torch.randn(cout, k, k, cin, device=dev) * 0.02
        dy = torch.randn(n, ho, wo, cout, device=dev) * 1e-3
        if zero:
            x.zero_(), wt.zero_(), dy.zero_()
        y = torch.empty(n, ho, wo, cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.zeros_like(wt)
        nw, pw = wt.numel(), ops.x3_plane(wt.numel())
        wh = torch.empty(3 * pw, device=dev, dtype=torch.int16)
        wth = torch.empty(3 * pw, device=dev, dtype=torch.int16)
        wtf = torch.empty(nw, device=dev)
        _lib.check(lib.hn_weight_flip_transpose_h2(P(wt), P(wh), P(wtf), P(wth), cout, k, k, cin, stride, pad, dil, s), "flip")
        xs = torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev)
        gs = torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev)
        _lib.check(lib.hn_absmax_scale(P(x), cin, n * h * w, cin, P(xs), s), "absmax")
        _lib.check(lib.hn_absmax_scale(P(dy), cout, n * ho * wo, cout, P(gs), s), "absmax")
        xplane, gplane = n * h * w * cin, n * ho * wo * cout
        xh = torch.zeros(2 * xplane, device=dev, dtype=torch.int16)
        gh = torch.zeros(2 * gplane, device=dev, dtype=torch.int16)
        _lib.check(lib.hn_tensor_to_planes(P(x), cin, n * h * w, cin, P(xs), P(xh), xplane, cin, s), "planes")
        _lib.check(lib.hn_tensor_to_planes(P(dy), cout, n * ho * wo, cout, P(gs), P(gh), gplane, cout, s), "planes")
        d = _lib.HnConvDesc(n, h, w, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
        flops = 2.0 * n * ho * wo * cout * k * k * cin
        tf = lambda ms: flops / ms / 1e9
        line = ["%-28s" % sh]
        if "fwd" in kinds:
            res = {}
            for name, v in old.items():
                if lib.hn_conv2d_fwd_scaled(P(x), P(xs), P(wt), P(wh), None, P(y), None, ctypes.byref(d), 0, 0.0, v, s) == 0:
                    res[name] = timeit(lambda: lib.hn_conv2d_fwd_scaled(P(x), P(xs), P(wt), P(wh), None, P(y), None, ctypes.byref(d), 0, 0.0, v, s))
            new = {}
            for tile in (0, 1, 2, 3, 4, 5, 6):
                if tile == 0 and cout <= 64:
                    continue
                if lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), None, P(y), None, None, ctypes.byref(d), 0, 0.0, tile, s) == 0:
                    new[tile] = timeit(lambda: lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), None, P(y), None, None, ctypes.byref(d), 0, 0.0, tile, s))
            line.append("fwd scaled " + " ".join("%s %.0f" % (k2, tf(v2)) for k2, v2 in res.items()) +
                        " | planes " + " ".join("t%d %.0f" % (k2, tf(v2)) for k2, v2 in new.items()) +
                        " | best %.2fx" % (min(res.values()) / min(new.values())))
        if "dgrad" in kinds:
            res = {}
            for name, v in old.items():
                if lib.hn_conv2d_dgrad_scaled(P(dy), P(gs), P(wtf), P(wth), P(dx), None, 0, ctypes.byref(d), v, s) == 0:
                    res[name] = timeit(lambda: lib.hn_conv2d_dgrad_scaled(P(dy), P(gs), P(wtf), P(wth), P(dx), None, 0, ctypes.byref(d), v, s))
            new = {}
            for tile in (0, 1, 2, 3, 4, 5, 6):
                if tile == 0 and cin <= 64:
                    continue
                if lib.hn_conv2d_dgrad_planes(P(gh), gplane, P(gs), P(wth), P(dx), None, 0, ctypes.byref(d), tile, s) == 0:
                    new[tile] = timeit(lambda: lib.hn_conv2d_dgrad_planes(P(gh), gplane, P(gs), P(wth), P(dx), None, 0, ctypes.byref(d), tile, s))
            line.append("dgrad scaled " + " ".join("%s %.0f" % (k2, tf(v2)) for k2, v2 in res.items()) +
                        " | planes " + " ".join("t%d %.0f" % (k2, tf(v2)) for k2, v2 in new.items()) +
                        " | best %.2fx" % (min(res.values()) / min(new.values())))
        if "wgrad" in kinds and hasattr(lib, "hn_conv2d_wgrad_planes"):
            res, new = {}, {}
            for sp in (0, -2, -3):
                if lib.hn_conv2d_wgrad_scaled(P(x), P(xs), P(dy), P(gs), P(dw), ctypes.byref(d), sp, 0, 0, s) == 0:
                    res[sp] = timeit(lambda: lib.hn_conv2d_wgrad_scaled(P(x), P(xs), P(dy), P(gs), P(dw), ctypes.byref(d), sp, 0, 0, s))
                for var in (0, 2, 4 + 2, 16 + 2, 4 + 16 + 2):
                    if lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw), ctypes.byref(d), sp, var, 0, s) == 0:
                        new[(sp, var)] = timeit(lambda: lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw), ctypes.byref(d), sp, var, 0, s))
            if res and new:
                line.append("wgrad scaled " + " ".join("s%d %.0f" % (k2, tf(v2)) for k2, v2 in res.items()) +
                            " | planes " + " ".join("s%dv%d %.0f" % (k2[0], k2[1], tf(v2)) for k2, v2 in new.items()) +
                            " | best %.2fx" % (min(res.values()) / min(new.values())))
        print("\n   ".join(line), flush=True)


if __name__ == "__main__":
    main()
