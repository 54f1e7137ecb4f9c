"""BatchNorm kernel microbenchmark (GPU box): times hn_bn_act_fwd / bwd_reduce / bwd_apply and reports the HBM rate.

usage: python tools/bench_bn.py [N,H,W,C[,res] ...]     (res = 1: residual add + ReLU mask from z)
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heatnet_amd import ops  # noqa: E402

DEFAULT = ["8,40,88,1024,1", "8,40,88,256,0", "8,80,176,256,1", "8,80,176,64,0", "8,160,352,64,0", "8,40,88,2048,1"]


def timeit(fn, reps=10, rounds=5):
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
    dev = torch.device("cuda:0")
    print("%-22s %10s %10s %10s   (us, TB/s of compulsory bytes)" % ("N,H,W,C,res", "fwd", "bwd_reduce", "bwd_apply"))
    for sh in sys.argv[1:] or DEFAULT:
        n, h, w, c, res = [int(v) for v in sh.split(",")]
        cl = torch.channels_last
        y = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=cl).requires_grad_()
        r = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=cl).requires_grad_() if res else None
        gamma, beta = torch.ones(c, device=dev, requires_grad=True), torch.zeros(c, device=dev, requires_grad=True)
        rm, rv = torch.zeros(c, device=dev), torch.ones(c, device=dev)
        ws = ops.BnWorkspace(c, dev)
        gz = torch.randn(n, c, h, w, device=dev).contiguous(memory_format=cl)
        nbytes = 4.0 * n * c * h * w
        t_f = timeit(lambda: ops.bn_act(y, None, gamma, beta, rm, rv, residual=r, training=True, act=ops.ACT_RELU, ws=ws))
        z = ops.bn_act(y, None, gamma, beta, rm, rv, residual=r, training=True, act=ops.ACT_RELU, ws=ws)
        t_b = timeit(lambda: torch.autograd.grad(z, [y] + ([r] if res else []), gz, retain_graph=True))
        # forward = channel_stats (1 read) + finalize + fwd (1-2 reads, 1 write); backward = reduce + apply
        f_bytes = nbytes * (1 + (3 if res else 2))
        b_bytes = nbytes * ((3 if res else 2) + (3 if res else 2) + (2 if res else 1))
        print("%-22s %10.1f %10s %10.1f   fwd(all) %.2f TB/s  bwd(reduce+apply) %.2f TB/s"
              % (sh, t_f * 1e3, "-", t_b * 1e3, f_bytes / t_f / 1e9, b_bytes / t_b / 1e9))


if __name__ == "__main__":
    main()
