"""HN_TAP_INNER=1 python tools/probes/tap_order_check.py: the plane-tensor convolution with the taps-inside-a-chunk K order against
the fp32-input f16x2 kernel (channels innermost): same products, another order of the fp32 accumulation -- prints the largest
difference relative to the largest output."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import test_gpu_planes as tp
from test_gpu_planes import P, rnd, make_slot, to_planes, _stream
from heatnet_amd import _lib, ops
import torch.nn.functional as F
lib = _lib.load(); dev = torch.device("cuda:0"); s = _stream()
for case in [(2, 256, 32, 40, 256, 3, 1, 2, 2), (1, 512, 24, 40, 512, 3, 1, 4, 4), (2, 128, 40, 44, 128, 3, 2, 1, 1), (4, 64, 96, 100, 64, 3, 1, 1, 1)]:
    for tile in (0, 2, 7):
        n, cin, h, w_, cout, k, stride, pad, dil = case
        ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w_, k, stride, pad, dil)
        x = F.relu(rnd(n, h, w_, cin, seed=11)).to(dev)
        wt = rnd(cout, k, k, cin, seed=12, scale=1.0 / np.sqrt(cin * k * k)).to(dev)
        b = rnd(cout, seed=13).to(dev)
        gy = (rnd(n, ho, wo, cout, seed=14) * 1e-3).to(dev)
        skip = (rnd(n, h, w_, cin, seed=15) * 1e-3).to(dev)
        nw, plane_w = wt.numel(), ops.x3_plane(wt.numel())
        wh, wth, wtf = (torch.empty(3 * plane_w, device=dev, dtype=torch.int16), torch.empty(3 * plane_w, device=dev, dtype=torch.int16), torch.empty(nw, device=dev))
        _lib.check(lib.hn_weight_flip_transpose_h2(P(wt), P(wh), P(wtf), P(wth), cout, k, k, cin, stride, pad, dil, s), "flip")
        xs, gs = make_slot(lib, x), make_slot(lib, gy)
        d = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
        lib.hn_set_deterministic(1)
        y0 = torch.empty(n, ho, wo, cout, device=dev)
        _lib.check(lib.hn_conv2d_fwd_scaled(P(x), P(xs), P(wt), P(wh), P(b), P(y0), None, ctypes.byref(d), 0, 0.0, 0, s), "fwd")
        dx0 = torch.empty_like(x)
        _lib.check(lib.hn_conv2d_dgrad_scaled(P(gy), P(gs), P(wtf), P(wth), P(dx0), P(skip), cin, ctypes.byref(d), 0, s), "dgrad")
        lib.hn_set_deterministic(0)
        xh, xplane = to_planes(lib, x, xs); gh, gplane = to_planes(lib, gy, gs)
        y1 = torch.full((n, ho, wo, cout), float("nan"), device=dev)
        _lib.check(lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), P(b), P(y1), None, None, ctypes.byref(d), 0, 0.0, tile, s), "fwd_planes")
        dx1 = torch.full_like(x, float("nan"))
        _lib.check(lib.hn_conv2d_dgrad_planes(P(gh), gplane, P(gs), P(wth), P(dx1), P(skip), cin, ctypes.byref(d), tile, s), "dgrad_planes")
        torch.cuda.synchronize()
        print(case, tile, "fwd", float((y1 - y0).abs().max() / y0.abs().max()), "dgrad", float((dx1 - dx0).abs().max() / dx0.abs().max()), flush=True)
