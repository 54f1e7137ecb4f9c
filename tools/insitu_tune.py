"""Second tuning pass INSIDE the training step (GPU box): the per-shape tuner (ops._pick) times every kernel candidate alone,
back to back with evicted caches; inside a step the same launch runs beside another stream's kernels, behind a producer that
left its operands in the caches, at the clock the surrounding kernels leave.  This tool takes the choices of the first pass,
then walks the geometries in the order of their time share and re-decides each one by the WHOLE step's time (coordinate
descent: a candidate is kept only if two measurements agree that the step got shorter), and writes the tune cache.

usage: python tools/insitu_tune.py [--workload seg] [--steps 12] [--top 60] [--min-gain-us 40]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="seg")
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--top", type=int, default=60)
    ap.add_argument("--min-gain-us", type=float, default=40.0)
    ap.add_argument("--height", type=int, default=bench.H)
    ap.add_argument("--width", type=int, default=bench.W)
    args = ap.parse_args()
    from heatnet_amd import optim, ops
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if args.workload == "seg_bf16":
        ops.set_conv_math("bf16")
    ops._candidates_seen = {}
    model = bench.build(args.workload, dev)
    arenas = optim.arenas_for(model)
    opt = optim.FlatRMSprop(arenas, lr=1e-4)
    batch = bench.synthetic_batch(0, dev, bench.BS, args.height, args.width)
    step = bench.make_step(args.workload, model, opt, batch, None)
    for _ in range(4):
        step()
    torch.cuda.synchronize()

    def step_ms(n=args.steps):
        step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    # time share of every geometry: the instrumented pass of bench.py
    records = []
    ops.set_conv_profiler(records)
    step()
    torch.cuda.synchronize()
    ops.set_conv_profiler(None)
    share = {}
    for rec in records:
        kind, e0, e1, shape = rec[0], rec[2], rec[3], rec[4]
        share[(kind,) + tuple(shape)] = share.get((kind,) + tuple(shape), 0.0) + e0.elapsed_time(e1)
    names = [f for f, _ in ops.HnConvDesc._fields_]
    seen = ops._candidates_seen
    keys = []
    for key, cands in seen.items():
        if len(cands) < 2 or key not in ops._variant_cache:
            continue
        d = dict(zip(names, key[1:]))
        sig = (key[0].split(".")[0], d["N"], d["Hin"], d["Win"], d["Cin"], d["Hout"], d["Wout"], d["Cout"], d["KH"], d["stride"], d["dil"])
        keys.append((share.get(sig, 0.0), key))
    keys.sort(reverse=True)
    keys = keys[:args.top]
    base = min(step_ms(), step_ms())
    print("baseline %.3f ms per step; %d geometries with alternatives, walking the top %d" % (base, len(seen), len(keys)), flush=True)
    changed = 0
    for ms, key in keys:
        cur = ops._variant_cache[key]
        best, best_t = cur, base
        for cand in seen[key]:
            if cand == cur:
                continue
            ops._variant_cache[key] = cand
            t1 = step_ms()
            if t1 < best_t - args.min_gain_us * 1e-3:
                t2 = step_ms()  # confirm
                if t2 < best_t - args.min_gain_us * 1e-3:
                    best, best_t = cand, max(t1, t2)
        ops._variant_cache[key] = best
        if best != cur:
            changed += 1
            ops._tune_dirty = True
            print("  %s %s: %s -> %s, step %.3f -> %.3f ms (its launches: %.3f ms per step)" % (key[0], key[1:], cur, best, base, best_t, ms),
                  flush=True)
            base = min(best_t, step_ms())
    final = min(step_ms(), step_ms())
    print("changed %d of %d; step %.3f ms" % (changed, len(keys), final), flush=True)
    ops._save_tune_cache()


if __name__ == "__main__":
    main()
