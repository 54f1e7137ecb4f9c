"""Throughput of the GPU augmentation pass (hn_thermal_augment) on loader-sized samples, next to the CPU oracle.

usage: python tools/bench_augment.py [batch]      (320x700 sources -> 320x640 crops... the loader's FR geometry)"""
import os
import random
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heatnet_amd import augment  # noqa: E402
from oracle import thermal_aug_ref as A  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    (h0, w0), (h, w) = (320, 700), (320, 640)
    dev = torch.device("cuda:0")
    r = np.random.RandomState(0)
    rgb = r.randint(0, 256, (n, h0, w0, 3)).astype(np.uint8)
    ir = r.randint(20000, 27000, (n, h0, w0)).astype(np.uint16)
    lab = r.randint(0, 13, (n, h0, w0)).astype(np.uint8)
    params = augment.draw_params(n, (h0, w0), (h, w), random.Random(1))
    for p in params:
        p["angle"] = p["angle"] if p["angle"] is not None else 7.0  # worst case: every sample rotated
    d_rgb, d_ir, d_lab = (torch.from_numpy(t).to(dev) for t in (rgb, ir, lab))
    for _ in range(3):
        augment.augment_batch(d_rgb, d_ir, d_lab, (h, w), params, drop_target="rgb")
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        augment.augment_batch(d_rgb, d_ir, d_lab, (h, w), params, drop_target="rgb")
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    out_bytes = n * h * w * (16 + 8)          # 4 fp32 channels + int64 label written
    in_bytes = n * h * w * 6 * 4              # 6 source bytes per tap, 4 taps (mostly cache hits)
    print("GPU: %d samples in %.3f ms = %.0f samples/s; written %.1f GB/s (+ up to %.1f GB/s of tap reads)"
          % (n, ms, n / ms * 1e3, out_bytes / ms / 1e6, in_bytes / ms / 1e6))
    t0 = time.time()
    k = min(n, 4)
    for i in range(k):
        p = params[i]
        A.augment_sample(rgb[i], ir[i], lab[i], p["crop"] + (h, w), p["flip"], p["angle"], drop=p["drop"], drop_rgb=True)
    dt = (time.time() - t0) / k
    print("CPU oracle (numpy, 1 core): %.1f ms per sample = %.1f samples/s" % (dt * 1e3, 1.0 / dt))


if __name__ == "__main__":
    main()
