"""Conv kernel microbenchmark (GPU box): times hn_conv2d_fwd / dgrad / wgrad for given geometries.

usage: python tools/bench_conv.py [shape ...]   shape = N,H,W,Cin,Cout,K,stride,dil   (pad = dil*(K-1)//2)
"""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heatnet_amd import _lib, ops  # noqa: E402

DEFAULT = ["8,40,88,10240,1024,1,1,1", "8,40,88,1024,2048,1,1,1", "8,40,88,1024,256,1,1,1", "8,40,88,256,1024,1,1,1",
           "8,40,88,512,512,3,1,4", "8,80,176,1024,256,3,1,1", "64,40,88,1024,1024,1,1,1", "8,80,176,64,256,1,1,1",
           "8,80,176,64,64,3,1,1", "8,320,704,64,64,3,1,1", "8,40,88,256,256,3,1,2"]


VAR = int(os.environ.get("VARIANT", "0"))
SPLIT = int(os.environ.get("SPLIT", "0"))


def timeit(fn, reps=6, rounds=6):
    """Best of `rounds` (each the mean of `reps` back-to-back launches): robust against clock / tenancy noise."""
    for _ in range(3):
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


def main():
    lib = _lib.load()
    dev = torch.device("cuda:0")
    shapes = sys.argv[1:] or DEFAULT
    kinds = os.environ.get("KINDS", "fwd,dgrad,wgrad").split(",")
    print("%-34s %8s %10s %10s %10s" % ("N,H,W,Cin,Cout,K,s,d", "GFLOP", "fwd TF/s", "dgrad TF/s", "wgrad TF/s"))
    for sh in shapes:
        n, h, w, cin, cout, k, stride, dil = [int(v) for v in sh.split(",")]
        pad = dil * (k - 1) // 2
        ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w, k, stride, pad, dil)
        x = torch.randn(n, h, w, cin, device=dev)
        wt = torch.randn(cout, k, k, cin, device=dev) * 0.02
        if os.environ.get("ZERO") == "1":  # data-dependent power: all-zero operands toggle no multiplier bits
            x.zero_()
            wt.zero_()
        y = torch.empty(n, ho, wo, cout, device=dev)
        dy = torch.randn(n, ho, wo, cout, device=dev)
        dx = torch.empty_like(x)
        dw = torch.empty_like(wt)
        wflip = torch.empty(cin * k * k * cout, device=dev)
        nw = cin * k * k * cout
        wx3 = torch.empty(3 * ops.x3_plane(nw), device=dev, dtype=torch.int16)
        wflipx3 = torch.empty(3 * ops.x3_plane(nw), device=dev, dtype=torch.int16)
        d = _lib.HnConvDesc(n, h, w, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())
        _lib.check(lib.hn_weight_flip_transpose(P(wt), None, P(wflip), P(wflipx3), cout, k, k, cin, stride, pad, dil, s), "flip")
        _lib.check(lib.hn_split_bf16x3(P(wt), P(wx3), nw, s), "split")
        flops = 2.0 * n * ho * wo * cout * k * k * cin
        res = {}
        if os.environ.get("H2") == "1":  # f16x2 arithmetic: forward only (the data gradient is the same kernel)
            xs = torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev)
            wh2 = torch.zeros(3 * ops.x3_plane(nw), device=dev, dtype=torch.int16)
            _lib.check(lib.hn_split_f16x2(P(wt), P(wh2), nw, s), "split_f16x2")

            def run_h2():
                _lib.check(lib.hn_absmax_scale(P(x), cin, n * h * w, cin, P(xs), s), "absmax")
                _lib.check(lib.hn_conv2d_fwd_scaled(P(x), P(xs), P(wt), P(wh2), None, P(y), None, ctypes.byref(d), 0, 0.0, VAR, s), "fwd_scaled")
            t_all = timeit(run_h2)
            t_conv = timeit(lambda: _lib.check(lib.hn_conv2d_fwd_scaled(P(x), P(xs), P(wt), P(wh2), None, P(y), None, ctypes.byref(d), 0, 0.0, VAR, s), "fwd_scaled"))
            y2 = y.clone()
            _lib.check(lib.hn_conv2d_fwd(P(x), P(wt), P(wx3), None, P(y), None, ctypes.byref(d), 0, 0.0, 129, s), "fwd")
            err = float((y2 - y).abs().max()) / max(float(y.abs().max()), 1e-30)
            print("%-34s %8.1f  f16x2 conv %7.1f TF/s   with absmax %7.1f TF/s   max|diff to bf16x3|/max %.2e  scale %s" % (
                sh, flops / 1e9, flops / t_conv / 1e9, flops / t_all / 1e9, err, int(xs.view(torch.int32).max())))
            continue
        if "fwd" in kinds:
            res["fwd"] = timeit(lambda: _lib.check(lib.hn_conv2d_fwd(P(x), P(wt), P(wx3), None, P(y), None, ctypes.byref(d), 0, 0.0, VAR, s), "fwd"))
        if "dgrad" in kinds:
            res["dgrad"] = timeit(lambda: _lib.check(lib.hn_conv2d_dgrad(P(dy), P(wflip), P(wflipx3), P(dx), None, 0, ctypes.byref(d), VAR, s), "dgrad"))
        if "wgrad" in kinds:
            res["wgrad"] = timeit(lambda: _lib.check(lib.hn_conv2d_wgrad(P(x), P(dy), P(dw), ctypes.byref(d), SPLIT, VAR, 0, s), "wgrad"))
        cols = ["%10.1f" % (flops / res[k2] / 1e9) if k2 in res else "%10s" % "-" for k2 in ("fwd", "dgrad", "wgrad")]
        print("%-34s %8.1f %s" % (sh, flops / 1e9, " ".join(cols)))


if __name__ == "__main__":
    main()
