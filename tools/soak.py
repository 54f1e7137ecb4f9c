"""Soak (GPU box): N training steps of the headline workload under the default arithmetic -- loss finite and falling on a fixed
batch, allocator footprint flat after warm-up (scale slots, weight planes, workspaces do not accumulate).
usage: python tools/soak.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    dev = torch.device("cuda:0")
    from heatnet_amd import ops, optim, train_step
    model = bench.build("seg", dev)
    opt = optim.FlatRMSprop(optim.arenas_for(model), lr=1e-4)
    b = bench.synthetic_batch(0, dev)
    losses, mem = [], []
    for i in range(steps):
        loss = train_step.segmentation_step(model, opt, b["rgb_day"], b["ir_day"], b["label_day"])
        if i % 25 == 0 or i == steps - 1:
            losses.append((i, float(loss)))
            mem.append(torch.cuda.memory_allocated(dev) / 2 ** 20)
            print("step %4d  loss %.5f  allocated %.0f MiB  reserved %.0f MiB  absmax passes so far %d" % (
                i, losses[-1][1], mem[-1], torch.cuda.memory_reserved(dev) / 2 ** 20, ops.absmax_launches), flush=True)
    assert all(torch.isfinite(torch.tensor(v)) for _, v in losses)
    assert losses[-1][1] < losses[0][1], "the loss on a fixed batch must fall"
    assert max(mem[2:]) - min(mem[2:]) < 64, "allocator footprint drifts: %s" % mem
    print("soak ok: conv math %s, %d steps, loss %.4f -> %.4f" % (ops.CONV_MATH, steps, losses[0][1], losses[-1][1]))


if __name__ == "__main__":
    main()
