"""Where a tile of the all-LDS-DMA convolution kernel spends its time (GPU box; diagnostic build `make -C heatnet-pub_amd/csrc
STAMPS=1`, loaded through HEATNET_HIP_LIB): thread 0 of every workgroup records s_memtime at the phase boundaries of
conv_igemm4_kernel into a buffer no output depends on; this tool launches the forward (with batch statistics and range keys,
as the fused node does) or the data gradient of one layer shape on plane tensors and prints the phase durations.

usage: HEATNET_HIP_LIB=heatnet-pub_amd/lib/stamps/libheatnet_hip_stamps.so python tools/stamp_conv.py [fwd|dgrad] N,H,W,Cin,Cout,K,stride,dil [tile]
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from heatnet_amd import _lib, ops  # noqa: E402

PHASES = ["rowinfo+sync", "scales", "addr+prologue issue", "first chunk lands", "K loop", "last DMA drain",
          "epilogue (issue)", "stores complete"]


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def main():
    kind = sys.argv[1] if len(sys.argv) > 1 else "fwd"
    sh = sys.argv[2] if len(sys.argv) > 2 else "8,80,176,64,256,1,1,1"
    tile = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    lib = _lib.load()
    lib.hn_debug_set_stamps.argtypes = [ctypes.c_void_p]
    lib.hn_debug_set_stamps.restype = ctypes.c_int
    dev = torch.device("cuda:0")
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    n, h, w, cin, cout, k, stride, dil = [int(v) for v in sh.split(",")]
    pad = dil * (k - 1) // 2
    ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w, k, stride, pad, dil)
    x = torch.relu(torch.randn(n, h, w, cin, device=dev))
    wt = torch.randn(cout, k, k, cin, device=dev) * 0.02
    dy = torch.randn(n, ho, wo, cout, device=dev) * 1e-3
    y = torch.empty(n, ho, wo, cout, device=dev)
    dx = torch.empty_like(x)
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
    stats = torch.zeros(16 * 2 * cout, device=dev, dtype=torch.float64)
    keys = torch.zeros(16 * 2 * cout, device=dev, dtype=torch.int32)
    flush = torch.empty(768 << 20, device=dev, dtype=torch.uint8)
    stamps = torch.zeros(1 << 20, device=dev, dtype=torch.int64)

    def launch():
        if kind == "fwd":
            _lib.check(lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), None, P(y), P(stats), P(keys), ctypes.byref(d), 0, 0.0,
                                                tile, s), "fwd")
        else:
            _lib.check(lib.hn_conv2d_dgrad_planes(P(gh), gplane, P(gs), P(wth), P(dx), None, 0, ctypes.byref(d), tile, s), "dgrad")

    for _ in range(3):
        launch()
    torch.cuda.synchronize()
    flush.zero_()  # operands come from HBM, as inside a training step
    torch.cuda.synchronize()
    lib.hn_debug_set_stamps(P(stamps))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    launch()
    e1.record()
    torch.cuda.synchronize()
    lib.hn_debug_set_stamps(None)
    st = stamps.cpu().numpy().reshape(-1, 16)
    live = st[:, 0] != 0
    st = st[live]
    nb = st.shape[0]
    rt0, rt1 = st[:, 9], st[:, 10]
    span_us = (rt1.max() - rt0.min()) / 100.0  # s_memrealtime: 100 MHz
    cyc = (st[:, 8] - st[:, 0]).astype(np.float64)
    rt = (rt1 - rt0).astype(np.float64) / 100.0
    ghz = np.median(cyc / np.maximum(rt, 1e-3)) / 1e3
    print("%s %s tile %d: %d workgroups, launch %.1f us by events, %.1f us first-start to last-end; clock %.2f GHz"
          % (kind, sh, tile, nb, e0.elapsed_time(e1) * 1e3, span_us, ghz))
    print("per workgroup: median %.2f us, p90 %.2f us" % (np.median(rt), np.percentile(rt, 90)))
    for i, name in enumerate(PHASES):
        dtc = (st[:, i + 1] - st[:, i]).astype(np.float64) / (ghz * 1e3)
        print("  %-22s median %6.2f us   p10 %6.2f   p90 %6.2f" % (name, np.median(dtc), np.percentile(dtc, 10), np.percentile(dtc, 90)))
    # inside the fourth K step of thread 0's wave (layers with at least four): the counted wait for its own DMAs of the chunk, then
    # the barrier -- i.e. the wait for the other waves' pieces of it
    ok = (st[:, 13] != 0) & (st[:, 15] != 0)
    if ok.any():
        for name, a, b in (("step 3: counted wait (own DMAs)", 13, 14), ("step 3: barrier", 14, 15)):
            dtc = (st[ok, b] - st[ok, a]).astype(np.float64) / (ghz * 1e3)
            print("  %-32s median %6.3f us   p10 %6.3f   p90 %6.3f" % (name, np.median(dtc), np.percentile(dtc, 10), np.percentile(dtc, 90)))
    # how the workgroups are spread in time: start offsets relative to the first
    start = (rt0 - rt0.min()) / 100.0
    end = (rt1 - rt0.min()) / 100.0
    qs = [0, 10, 25, 50, 75, 90, 100]
    print("  start offsets (us) at percentiles %s: %s" % (qs, " ".join("%.1f" % np.percentile(start, q) for q in qs)))
    print("  end offsets   (us) at percentiles %s: %s" % (qs, " ".join("%.1f" % np.percentile(end, q) for q in qs)))


if __name__ == "__main__":
    main()
