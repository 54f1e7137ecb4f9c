"""Diagnostic (GPU box): host time to ENQUEUE one training step vs GPU time to run it."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    from heatnet_amd import optim
    model = bench.build("seg", dev)
    arenas = optim.arenas_for(model)
    opt = optim.FlatRMSprop(arenas, lr=1e-4)
    batch = bench.synthetic_batch(0, dev)
    step = bench.make_step("seg", model, opt, batch, None)
    for _ in range(4):
        step()
    torch.cuda.synchronize()
    n = 8
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    t_host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    t_all = (time.perf_counter() - t0) / n
    print("host enqueue %.1f ms/step, wall %.1f ms/step (GPU-bound if host << wall)" % (1e3 * t_host, 1e3 * t_all))
    # host-only cost: same loop while the GPU queue is empty at start of each step
    ts = []
    for _ in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        step()
        ts.append(time.perf_counter() - t0)
    print("host enqueue from an idle queue: %.1f ms/step" % (1e3 * min(ts)))


if __name__ == "__main__":
    main()
