"""Plane tensors (f16x2 operands made by the producing pass) and the all-LDS-DMA convolution kernels that read them
(csrc/conv_igemm4.hip, csrc/conv_wgrad4.hip), through the C ABI.

With planes made with the SAME scale slot the kernels must reproduce the *_scaled entry points (which split the fp32 tensor in
registers) bit for bit: same LDS contents, same MFMA order.  The reference layers these kernels serve:
cm/models/extractors.py:84-102 (bottleneck), cm/models/pspnet.py:37-40 (decoder convolutions)."""
import ctypes

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def P(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_planes(lib, x_nhwc, slot, ldp=None):
    """[N, H, W, C] fp32 -> int16 buffer of 2 planes with the scale of `slot`."""
    from heatnet_amd import _lib
    n, h, w, c = x_nhwc.shape
    ldp = c if ldp is None else ldp
    npix = n * h * w
    plane = npix * ldp
    buf = torch.zeros(2 * plane, device=x_nhwc.device, dtype=torch.int16)
    _lib.check(lib.hn_tensor_to_planes(P(x_nhwc), c, npix, c, P(slot), P(buf), plane, ldp, _stream()), "to_planes")
    return buf, plane


def make_slot(lib, t_nhwc):
    from heatnet_amd import _lib, ops
    n, h, w, c = t_nhwc.shape
    slot = torch.zeros(ops.SCALE_SLOT_FLOATS, device=t_nhwc.device)
    _lib.check(lib.hn_absmax_scale(P(t_nhwc), c, n * h * w, c, P(slot), _stream()), "absmax")
    return slot


def test_plane_tensor_round_trip(dev):
    """(hi + lo) / s recovers the tensor to 2^-22 of each element (elements within 2^-18 of the maximum), exactly for fp16-
    representable values; scalar path (C % 4 != 0) agrees with the vector path."""
    from heatnet_amd import _lib
    lib = _lib.load()
    for c in (64, 6):
        x = (rnd(2, 9, 11, c, seed=1) * 3.0).to(dev)
        slot = make_slot(lib, x)
        buf, plane = to_planes(lib, x, slot)
        back = torch.empty_like(x)
        _lib.check(lib.hn_planes_to_tensor(P(buf), plane, c, P(slot), P(back), c, x.numel() // c, c, _stream()), "from_planes")
        err = ((back - x).abs() / x.abs().clamp_min(1e-30)).max().item()
        assert err <= 2.0 ** -21, err
        xi = torch.randint(-2048, 2048, x.shape, device=dev).float()  # exactly representable after any power-of-two scale
        slot = make_slot(lib, xi)
        buf, plane = to_planes(lib, xi, slot)
        _lib.check(lib.hn_planes_to_tensor(P(buf), plane, c, P(slot), P(back), c, x.numel() // c, c, _stream()), "from_planes")
        assert torch.equal(back, xi)


PLANE_CONV_CASES = [  # n, cin, h, w, cout, k, stride, pad, dil
    (2, 64, 40, 44, 256, 1, 1, 0, 1),
    (2, 256, 32, 40, 256, 3, 1, 2, 2),
    (2, 128, 33, 47, 320, 3, 1, 1, 1),    # ragged rows and columns
    (2, 128, 40, 44, 64, 3, 1, 1, 1),     # 64 output channels
    (2, 128, 40, 44, 128, 3, 2, 1, 1),    # stride 2 (forward gather; data gradient by parity classes)
    (2, 256, 40, 44, 512, 1, 2, 0, 1),    # strided 1x1 (downsample)
    (1, 512, 24, 40, 512, 3, 1, 4, 4),    # dilation 4
    (3, 96, 30, 30, 160, 1, 1, 0, 1),     # Cin % 64 != 0, Cout % 64 != 0
    (4, 64, 80, 88, 256, 1, 1, 0, 1),     # several rounds of workgroups
    (4, 64, 96, 100, 64, 3, 1, 1, 1),     # the same with a 3x3 filter, ragged last row tile, narrow output
]


def _same_conv(got, ref, taps):
    """1x1 filters: bit for bit.  Larger filters: conv_igemm4_kernel walks the K dimension taps-inside-a-32-channel-chunk, the
    fp32-input kernels channels-inside-a-tap -- the same products summed in fp32 in another order (measured <= 3e-6 of the
    largest output); the tiles of conv_igemm4_kernel agree among themselves bit for bit (checked by the callers)."""
    if taps == 1:
        return torch.equal(got, ref)
    return float((got - ref).abs().max()) <= 6e-6 * float(ref.abs().max())


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("case", PLANE_CONV_CASES, ids=str)
def test_conv_planes_forward_dgrad_bitwise(dev, case, tile):
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    s = _stream()
    n, cin, h, w_, cout, k, stride, pad, dil = case
    ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w_, k, stride, pad, dil)
    x = F.relu(rnd(n, h, w_, cin, seed=11)).to(dev)
    wt = rnd(cout, k, k, cin, seed=12, scale=1.0 / np.sqrt(cin * k * k)).to(dev)
    b = rnd(cout, seed=13).to(dev)
    gy = (rnd(n, ho, wo, cout, seed=14) * 1e-3).to(dev)
    skip = (rnd(n, h, w_, cin, seed=15) * 1e-3).to(dev)
    nw, plane_w = wt.numel(), ops.x3_plane(wt.numel())
    wh, wth, wtf = (torch.empty(3 * plane_w, device=dev, dtype=torch.int16), torch.empty(3 * plane_w, device=dev, dtype=torch.int16),
                    torch.empty(nw, device=dev))
    _lib.check(lib.hn_weight_flip_transpose_h2(P(wt), P(wh), P(wtf), P(wth), cout, k, k, cin, stride, pad, dil, s), "flip")
    xs, gs = make_slot(lib, x), make_slot(lib, gy)
    d = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
    # reference: the fp32-input f16x2 kernels (deterministic: no split-K atomics)
    lib.hn_set_deterministic(1)
    try:
        y0 = torch.empty(n, ho, wo, cout, device=dev)
        st0 = torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.float64)
        _lib.check(lib.hn_conv2d_fwd_scaled(P(x), P(xs), P(wt), P(wh), P(b), P(y0), P(st0), ctypes.byref(d), 0, 0.0, 0, s), "fwd")
        dx0 = torch.empty_like(x)
        _lib.check(lib.hn_conv2d_dgrad_scaled(P(gy), P(gs), P(wtf), P(wth), P(dx0), P(skip), cin, ctypes.byref(d), 0, s), "dgrad")
    finally:
        lib.hn_set_deterministic(0)
    xh, xplane = to_planes(lib, x, xs)
    gh, gplane = to_planes(lib, gy, gs)
    y1 = torch.full((n, ho, wo, cout), float("nan"), device=dev)
    st1 = torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.float64)
    keys = torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.int32)
    _lib.check(lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), P(b), P(y1), P(st1), P(keys), ctypes.byref(d), 0, 0.0, tile, s),
               "fwd_planes")
    assert _same_conv(y1, y0, k * k)
    if k > 1 and tile != 0:  # the tiles among themselves: bit for bit
        yt = torch.full_like(y1, float("nan"))
        _lib.check(lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), P(b), P(yt), None, None, ctypes.byref(d), 0, 0.0, 0, s), "tile 0")
        assert torch.equal(yt, y1)
    # statistics: per-lane fp32 partial sums over a tile's rows, then double -- the grouping depends on the tile
    a1 = st1.view(ops.STAT_REPLICAS, 2, cout).sum(0)
    yd = y1.double().reshape(-1, cout)
    assert float((a1[0] - yd.sum(0)).abs().max()) <= 1e-5 * float(yd.abs().sum(0).max())
    assert float((a1[1] - (yd * yd).sum(0)).abs().max()) <= 1e-5 * float((yd * yd).sum(0).max())
    # range keys: exact per-channel max / min of y
    kk = keys.view(ops.STAT_REPLICAS, 2, cout).cpu().numpy().astype(np.uint32).max(0)

    def unkey(u):
        u = u.astype(np.uint32)
        b_ = np.where(u & 0x80000000, u ^ np.uint32(0x80000000), ~u)
        return b_.astype(np.uint32).view(np.float32)
    ymax, ymin = unkey(kk[0]), -unkey(kk[1])
    yc = y1.reshape(-1, cout).cpu().numpy()
    assert np.array_equal(ymax, yc.max(0)) and np.array_equal(ymin, yc.min(0))
    dx1 = torch.full_like(x, float("nan"))
    _lib.check(lib.hn_conv2d_dgrad_planes(P(gh), gplane, P(gs), P(wth), P(dx1), P(skip), cin, ctypes.byref(d), tile, s), "dgrad_planes")
    assert _same_conv(dx1, dx0, k * k)
    if k > 1 and tile != 0:
        dxt = torch.full_like(x, float("nan"))
        _lib.check(lib.hn_conv2d_dgrad_planes(P(gh), gplane, P(gs), P(wth), P(dxt), P(skip), cin, ctypes.byref(d), 0, s), "tile 0")
        assert torch.equal(dxt, dx1)
    # fused ReLU epilogue
    y2 = torch.empty_like(y1)
    _lib.check(lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), P(b), P(y2), None, None, ctypes.byref(d), 1, 0.0, tile, s), "relu")
    assert torch.equal(y2, torch.relu(y1))


def test_conv_planes_refusals(dev):
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    s = _stream()
    n, cin, h, w_, cout, k = 2, 64, 16, 24, 64, 3
    x = rnd(n, h, w_, cin, seed=21).to(dev)
    wt = rnd(cout, k, k, cin, seed=22).to(dev)
    plane_w = ops.x3_plane(wt.numel())
    wh = torch.empty(3 * plane_w, device=dev, dtype=torch.int16)
    _lib.check(lib.hn_split_f16x2(P(wt), P(wh), wt.numel(), s), "split")
    xs = make_slot(lib, x)
    xh, xplane = to_planes(lib, x, xs)
    y = torch.empty(n, h, w_, cout, device=dev)
    d = _lib.HnConvDesc(n, h, w_, cin, cin, h, w_, cout, cout, k, k, 1, 1, 1)
    # (x_scale = NULL is the one-plane bf16 form, test_conv_planes_bf16_bitwise)
    assert lib.hn_conv2d_fwd_planes(None, xplane, P(xs), P(wh), None, P(y), None, None, ctypes.byref(d), 0, 0.0, 0, s) == -1
    d48 = _lib.HnConvDesc(n, h, w_, 48, 48, h, w_, cout, cout, k, k, 1, 1, 1)  # Cin % 32 != 0
    assert lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), None, P(y), None, None, ctypes.byref(d48), 0, 0.0, 0, s) == -4
    keys = torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.int32)  # range keys need the statistics
    assert lib.hn_conv2d_fwd_planes(P(xh), xplane, P(xs), P(wh), None, P(y), None, P(keys), ctypes.byref(d), 0, 0.0, 0, s) == -1


WGRAD_CASES = [  # n, cin, h, w, cout, k, stride, pad, dil
    (2, 64, 40, 44, 256, 1, 1, 0, 1),
    (2, 256, 32, 40, 256, 3, 1, 2, 2),
    (2, 128, 33, 47, 320, 3, 1, 1, 1),    # ragged pixel count, Cout = 5 x 64
    (2, 128, 40, 44, 64, 3, 1, 1, 1),
    (2, 128, 40, 44, 128, 3, 2, 1, 1),    # stride 2
    (2, 256, 40, 44, 512, 1, 2, 0, 1),
    (1, 512, 24, 40, 512, 3, 1, 4, 4),
    (3, 192, 30, 30, 192, 1, 1, 0, 1),    # 64-wide tiles both ways
]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=str)
def test_wgrad_planes_bitwise(dev, case):
    """hn_conv2d_wgrad_planes against hn_conv2d_wgrad_scaled: bitwise with one reduction split and through the deterministic
    (slab) form, to float-atomic noise with the default split; the accumulate flag; every tile choice."""
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    s = _stream()
    n, cin, h, w_, cout, k, stride, pad, dil = case
    ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w_, k, stride, pad, dil)
    x = F.relu(rnd(n, h, w_, cin, seed=31)).to(dev)
    gy = (rnd(n, ho, wo, cout, seed=32) * 1e-3).to(dev)
    xs, gs = make_slot(lib, x), make_slot(lib, gy)
    xh, xplane = to_planes(lib, x, xs)
    gh, gplane = to_planes(lib, gy, gs)
    d = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
    dw0 = torch.empty(cout, k, k, cin, device=dev)
    _lib.check(lib.hn_conv2d_wgrad_scaled(P(x), P(xs), P(gy), P(gs), P(dw0), ctypes.byref(d), 1, 0, 0, s), "wgrad")
    for tm, tn, bk in ((0, 0, 0), (1, 1, 0), (2, 1, 0), (1, 2, 0), (1, 1, 2), (2, 1, 2), (1, 2, 2)):  # bk 2: three LDS stages
        var = 4 * tm + 16 * tn
        dw0v = torch.empty_like(dw0)
        _lib.check(lib.hn_conv2d_wgrad_scaled(P(x), P(xs), P(gy), P(gs), P(dw0v), ctypes.byref(d), 1, var, 0, s), "wgrad")
        dw1 = torch.full_like(dw0, float("nan"))
        _lib.check(lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw1), ctypes.byref(d), 1, var + bk, 0, s),
                   "wgrad_planes")
        assert torch.equal(dw0v, dw1), (tm, tn)
        assert torch.equal(dw0, dw1)  # (tilings agree bitwise as well)
    if cin % 256 == 0 and cout % 256 == 0:  # the 256 x 256 eight-wave tile (variant 60)
        dw1 = torch.full_like(dw0, float("nan"))
        _lib.check(lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw1), ctypes.byref(d), 1, 60, 0, s),
                   "wgrad_planes 256 x 256")
        assert torch.equal(dw0, dw1)
    else:
        assert lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw1), ctypes.byref(d), 1, 60, 0, s) == -4
    # default split: float atomics
    dw2 = torch.empty_like(dw0)
    _lib.check(lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw2), ctypes.byref(d), 0, 0, 0, s), "split")
    assert float((dw2 - dw0).abs().max()) <= 2e-5 * float(dw0.abs().max())
    # accumulate into a zeroed / pre-filled dW
    dw3 = dw0.clone()
    _lib.check(lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw3), ctypes.byref(d), 0, 0, 2, s), "acc")
    assert float((dw3 - 2 * dw0).abs().max()) <= 4e-5 * float(dw0.abs().max())
    # deterministic form
    wsp = torch.empty(4 * dw0.numel(), device=dev)
    dw4, dw5 = torch.empty_like(dw0), torch.empty_like(dw0)
    _lib.check(lib.hn_conv2d_wgrad_det_scaled(P(x), P(xs), P(gy), P(gs), P(dw4), ctypes.byref(d), 4, 0, 0, P(wsp), s), "det")
    _lib.check(lib.hn_conv2d_wgrad_det_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw5), ctypes.byref(d), 4, 0, 0, P(wsp), s),
               "det_planes")
    assert torch.equal(dw4, dw5)
    # refusal: Cout % 64 != 0
    d32 = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, cout - 32, cout, k, k, stride, pad, dil)
    assert lib.hn_conv2d_wgrad_planes(P(xh), xplane, P(xs), P(gh), gplane, P(gs), P(dw1), ctypes.byref(d32), 0, 0, 0, s) == -4


FUSED_CASES = [  # cin, cout, k, stride, dil, act, residual ("f32" | "planes" | None), mask, out
    (64, 128, 1, 1, 1, "relu", None, False, "planes"),
    (128, 128, 3, 1, 2, "relu", None, False, "planes"),
    (128, 256, 1, 1, 1, "relu", "f32", False, "dual"),
    (128, 256, 1, 1, 1, "relu", "planes", False, "planes"),
    (128, 128, 3, 2, 1, "relu", None, False, "f32"),
    (64, 256, 1, 2, 1, "none", None, False, "f32"),
    (128, 64, 3, 1, 1, "prelu", None, True, "f32"),
]


@pytest.mark.parametrize("math", ["f16x2", "bf16", "bf16y"])
@pytest.mark.parametrize("case", FUSED_CASES, ids=str)
def test_fused_conv_bn_act_matches_unfused(dev, case, math):
    """layers.conv_bn_act through the fused plane-tensor node (conv_igemm4 + range-bounded scales + plane-writing BatchNorm
    passes + conv_wgrad4) against the same call with HEATNET_PLANES off (fp32 tensors, operands split in the kernels): outputs,
    input / residual / parameter gradients and running statistics.  Without a residual the bound equals the maximum, both
    paths use the same scales and the convolution outputs agree bitwise; with one the scale may differ by a factor of two.
    math = "bf16" (the mixed-precision arithmetic): the fused node keeps ONE bf16 image per operand; the unfused path rounds the
    same fp32 values to bf16 in the convolution loaders, so everything agrees to fp32 noise except what was STORED in bf16 --
    a plane output, a plane residual (2^-9 relative).  "bf16y": the node's default under bf16 -- it also keeps the
    pre-normalisation tensor y in bf16 (HN_Y_BF16; "bf16" above runs with HEATNET_BF16_Y off): z moves by 2^-9 |y| / sigma,
    the gradients are compared in the norm."""
    import torch.nn as nn
    from heatnet_amd import layers, ops
    y16 = math == "bf16y"
    math = "bf16" if y16 else math
    prev_y16, ops.BF16_Y = ops.BF16_Y, y16
    cin, cout, k, stride, dil, act, res_kind, use_mask, out_mode = case
    n, h, w_ = 2, 72, 80
    pad = dil * (k - 1) // 2
    ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w_, k, stride, pad, dil)
    acts = {"relu": ops.ACT_RELU, "prelu": ops.ACT_PRELU, "none": ops.ACT_NONE}
    torch.manual_seed(5)

    def run(planes):
        prev = ops.PLANES
        ops.PLANES = planes
        try:
            conv = layers.HnConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=(act == "prelu")).to(dev)
            bn = layers.HnBatchNorm2d(cout).to(dev)
            prelu = nn.Parameter(torch.full((1,), 0.25, device=dev)) if act == "prelu" else None
            g = torch.Generator().manual_seed(7)
            with torch.no_grad():
                conv.weight.copy_((torch.randn(conv.weight.shape, generator=g) / np.sqrt(cin * k * k)).to(dev))
                if conv.bias is not None:
                    conv.bias.copy_(torch.randn(cout, generator=g).to(dev))
                bn.weight.copy_((torch.rand(cout, generator=g) + 0.5).to(dev))
                bn.bias.copy_((torch.randn(cout, generator=g) * 0.3).to(dev))
            x = F.relu(torch.randn(n, cin, h, w_, generator=g)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            res = None
            res_leaf = None
            if res_kind is not None:
                res_leaf = torch.randn(n, cout, ho, wo, generator=g).abs().to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
                res = res_leaf * 1.0  # a non-leaf, as in the network
                if res_kind == "planes" and planes:
                    # what a preceding fused block hands over: a plane tensor
                    class _ToPlanes(torch.autograd.Function):
                        @staticmethod
                        def forward(ctx, t):
                            pt, slot = ops.planes_of(t, math == "bf16")
                            return pt.clone()

                        @staticmethod
                        def backward(ctx, g_):
                            return g_
                    pt, slot = ops.planes_of(res, math == "bf16")
                    res = _ToPlanes.apply(res)
                    res._hn_planes = slot
            mask = None
            if use_mask:
                mask = (torch.rand(n, cout, generator=g) > 0.3).float().to(dev) / 0.7
            z = layers.conv_bn_act(x * 1.0, conv, bn, acts[act], residual=res, prelu_weight=prelu, drop_mask=mask,
                                   out=out_mode, mask_max=1 / 0.7 if use_mask else 1.0)
            zf = ops.planes_to_tensor(z) if ops.is_planes(z) else z.detach()
            twin = getattr(z, "_hn_twin", None)
            if twin is not None:
                assert float((ops.planes_to_tensor(twin[1]) - z.detach()).abs().max()) <= \
                    (2.0 ** -8 if math == "bf16" else 2.0 ** -21) * float(z.detach().abs().max())
            gz = (torch.randn(n, cout, ho, wo, generator=g) * 1e-3).to(dev).contiguous(memory_format=torch.channels_last)
            z.backward(gz)
            outs = dict(z=zf, dx=x.grad, dw=conv.weight.grad, dgamma=bn.weight.grad, dbeta=bn.bias.grad,
                        rm=bn.running_mean.clone(), rv=bn.running_var.clone())
            if res_leaf is not None:
                outs["dres"] = res_leaf.grad
            if prelu is not None:
                outs["dprelu"] = prelu.grad
                outs["db"] = conv.bias.grad
            return {k_: v.detach().float().cpu() for k_, v in outs.items()}
        finally:
            ops.PLANES = prev
    prev_math = ops.set_conv_math(math)
    try:
        a, b = run(True), run(False)
    finally:
        ops.set_conv_math(prev_math)
        ops.BF16_Y = prev_y16
    stored16 = math == "bf16" and (out_mode == "planes" or res_kind == "planes")  # something was kept in bf16
    for key in b:
        ref = b[key]
        tol = 2e-6 if key in ("z", "rm", "rv") else 2e-5
        if key == "dprelu":
            tol = 1e-3  # one scalar: a sum of 10^5 cancelling terms
        if stored16 and key == "z":
            tol = 2.0 ** -8
        elif math == "bf16" and key not in ("z", "rm", "rv"):
            # gradients under bf16: the unfused path runs a few geometries (strided data gradients) on the fp32 kernels, i.e.
            # WITHOUT rounding dY to bf16, and the ReLU mask behind a plane residual sees the bf16-rounded residual
            tol = 4e-3
        if y16 and key == "z":
            tol = 2.0 ** -6  # y / sigma reaches ~5, max |z| is of that order too: a few bf16 steps of y
        if (y16 or (math == "bf16" and res_kind == "planes")) and key not in ("z", "rm", "rv"):
            # the ReLU behind the residual add sees the bf16-rounded residual: a few masks flip, and with them whole
            # gradient elements -- compared in the norm
            # (bf16 y: u = y * scale + shift of ~0.3 % of the elements changes sign with the rounding of y; against this test's
            # white-noise dz every flipped mask is a whole element of error -- sqrt(0.003) ~ 5 % in the norm)
            assert float((a[key] - ref).norm()) <= (8e-2 if y16 else 3e-2) * float(ref.norm()), key  # (db: both exactly zero)
            continue
        if k > 1 and key not in ("z", "rm", "rv"):
            # filters with taps: conv_igemm4_kernel sums the K dimension taps-inside-a-chunk, the fp32-input kernels of the
            # unfused path channels-inside-a-tap -- y agrees to fp32 rounding, not bit for bit, so the ReLU mask of an element with
            # u = y * scale + shift within a rounding of zero may flip, and with it one whole element of this test's white-noise
            # dz (seen: one element of 1.5 M, 6 % of the largest dx around it) -- compared in the norm
            assert float((a[key] - ref).norm()) <= 3e-3 * float(ref.norm()), key
            continue
        err = float((a[key] - ref).abs().max())
        assert err <= tol * max(float(ref.abs().max()), 1e-30), (key, err, float(ref.abs().max()))


def test_f16x2_range_guard_moves_heavy_tailed_layers_to_bf16x3(dev):
    """Per-ELEMENT accuracy on a heavy-tailed gradient operand, and the guard that acts on it.  dY of a 1x1 convolution has
    one magnitude per pixel, spread over 2^36; dX[p] depends on dY[p] alone, so each output row shows the precision its input
    row kept.  f16x2 (one scale per tensor): rows within 2^-18 of the maximum are fp32-accurate, rows far below it lose bits
    -- absolute error stays below max * 2^-36, relative error grows.  The census counts the lossy fraction; with the guard
    the layer moves to bf16x3, whose per-row relative error is fp32-grade for EVERY row."""
    import torch.nn as nn
    from heatnet_amd import ops
    n, cin, h, w_, cout = 2, 128, 40, 44, 128
    g = torch.Generator().manual_seed(3)
    x = torch.randn(n, cin, h, w_, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / np.sqrt(cin)
    expo = torch.linspace(0, -36, n * h * w_).reshape(n, 1, h, w_)         # per-pixel magnitude 2^0 ... 2^-36
    gy = torch.randn(n, cout, h, w_, generator=g) * torch.exp2(expo)
    ref = torch.einsum("nkhw,kc->nchw", gy.double(), wt.double()[:, :, 0, 0])  # dX in fp64

    def run():
        xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        y = ops.conv2d(xd, wparam, None, 1, 0, 1)
        y.backward(gy.to(dev).contiguous(memory_format=torch.channels_last))
        return xd.grad.detach().cpu().double()

    def row_rel(dx):
        num = (dx - ref).pow(2).sum(1).sqrt().reshape(-1)
        den = ref.pow(2).sum(1).sqrt().reshape(-1)
        return num / den

    prev_math = ops.set_conv_math("f16x2")
    prev_census = ops.set_range_census(True)
    try:
        wparam = nn.Parameter(wt.to(dev).contiguous(memory_format=torch.channels_last))
        dx = run()
        rel = row_rel(dx)
        mags = torch.exp2(expo).reshape(-1)
        near = mags >= 2.0 ** -14                      # comfortably inside the 2^-18 window (row norms vary by a few bits)
        assert float(rel[near].max()) < 2e-6
        assert float(rel[mags < 2.0 ** -30].max()) > 1e-4   # the documented loss: these rows keep < 10 bits
        assert float((dx - ref).abs().max()) < float(ref.abs().max()) * 2.0 ** -20
        rows = [r for r in ops.range_report() if r[5] is wparam]
        assert len(rows) == 1
        _, nz_a, frac_a, nz_g, frac_g, _ = rows[0]
        assert nz_g == gy.numel() and 0.4 < frac_g < 0.75   # over half the rows lie more than 2^18 below the maximum
        assert frac_a < 1e-3                                 # the Gaussian activation operand is fine
        moved = ops.range_guard_update(threshold=0.05)
        assert [m[0] for m in moved] == [tuple(wparam.shape)] and wparam._hn_no_f16x2
        rel2 = row_rel(run())                                # now on bf16x3
        assert float(rel2.max()) < 2e-6
    finally:
        ops.set_range_census(prev_census)
        ops.set_conv_math(prev_math)
        ops._census.clear()


@pytest.mark.parametrize("tile", [0, 1, 2, 3, 4, 5, 6, 7, 8])
@pytest.mark.parametrize("case", PLANE_CONV_CASES[:5] + PLANE_CONV_CASES[6:], ids=str)
def test_conv_planes_bf16_bitwise(dev, case, tile):
    """The mixed-precision bf16 arithmetic on ONE bf16 image per operand (x_scale = NULL): forward, data gradient and weight
    gradient equal, bit for bit, the kernels that round the fp32 operands to bf16 in their loaders (variant math 3)."""
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    s = _stream()
    n, cin, h, w_, cout, k, stride, pad, dil = case
    ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w_, k, stride, pad, dil)
    x = F.relu(rnd(n, h, w_, cin, seed=41)).to(dev)
    wt = rnd(cout, k, k, cin, seed=42, scale=1.0 / np.sqrt(cin * k * k)).to(dev)
    b = rnd(cout, seed=43).to(dev)
    gy = (rnd(n, ho, wo, cout, seed=44) * 1e-3).to(dev)
    skip = (rnd(n, h, w_, cin, seed=45) * 1e-3).to(dev)
    nw, plane_w = wt.numel(), ops.x3_plane(wt.numel())
    w3, wt3, wtf = (torch.empty(3 * plane_w, device=dev, dtype=torch.int16), torch.empty(3 * plane_w, device=dev, dtype=torch.int16),
                    torch.empty(nw, device=dev))
    _lib.check(lib.hn_weight_flip_transpose(P(wt), P(w3), P(wtf), P(wt3), cout, k, k, cin, stride, pad, dil, s), "flip")
    d = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
    vb = ops._variant(0, 0, 0, 3)
    lib.hn_set_deterministic(1)
    try:
        y0 = torch.empty(n, ho, wo, cout, device=dev)
        _lib.check(lib.hn_conv2d_fwd(P(x), P(wt), P(w3), P(b), P(y0), None, ctypes.byref(d), 0, 0.0, vb, s), "fwd")
        dx0 = torch.empty_like(x)
        _lib.check(lib.hn_conv2d_dgrad(P(gy), P(wtf), P(wt3), P(dx0), P(skip), cin, ctypes.byref(d), vb, s), "dgrad")
    finally:
        lib.hn_set_deterministic(0)
    xb = torch.zeros(n * h * w_ * cin, device=dev, dtype=torch.int16)
    gb = torch.zeros(n * ho * wo * cout, device=dev, dtype=torch.int16)
    _lib.check(lib.hn_tensor_to_planes(P(x), cin, n * h * w_, cin, None, P(xb), xb.numel(), cin, s), "to bf16")
    _lib.check(lib.hn_tensor_to_planes(P(gy), cout, n * ho * wo, cout, None, P(gb), gb.numel(), cout, s), "to bf16")
    assert torch.equal(xb.view(torch.bfloat16).float().view(n, h, w_, cin), x.to(torch.bfloat16).float())
    y1 = torch.full_like(y0, float("nan"))
    _lib.check(lib.hn_conv2d_fwd_planes(P(xb), xb.numel(), None, P(w3), P(b), P(y1), None, None, ctypes.byref(d), 0, 0.0, tile, s), "fwd")
    assert _same_conv(y1, y0, k * k)
    if k > 1 and tile != 0:
        yt = torch.full_like(y0, float("nan"))
        _lib.check(lib.hn_conv2d_fwd_planes(P(xb), xb.numel(), None, P(w3), P(b), P(yt), None, None, ctypes.byref(d), 0, 0.0, 0, s), "tile 0")
        assert torch.equal(yt, y1)
    # HN_Y_BF16: the same output rounded (to nearest even) to bf16 elements, the batch statistics still from the fp32 values
    y2 = torch.zeros(n, ho, wo, cout, device=dev, dtype=torch.bfloat16)
    st1, st2 = (torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.float64) for _ in range(2))
    _lib.check(lib.hn_conv2d_fwd_planes(P(xb), xb.numel(), None, P(w3), P(b), P(y1), P(st1), None, ctypes.byref(d), 0, 0.0, tile, s), "fwd")
    _lib.check(lib.hn_conv2d_fwd_planes(P(xb), xb.numel(), None, P(w3), P(b), P(y2), P(st2), None, ctypes.byref(d), ops.HN_Y_BF16, 0.0,
                                        tile, s), "fwd bf16 y")
    assert torch.equal(y2, y1.to(torch.bfloat16))
    assert torch.allclose(st1.view(16, -1).sum(0), st2.view(16, -1).sum(0), rtol=1e-11, atol=0)
    xs = torch.ones(64 * 16, device=dev)  # (a scale slot: the fp32-result arithmetic refuses the flag)
    assert lib.hn_conv2d_fwd_planes(P(xb), xb.numel(), P(xs), P(w3), P(b), P(y2), None, None, ctypes.byref(d), ops.HN_Y_BF16, 0.0,
                                    tile, s) == -1
    dx1 = torch.full_like(x, float("nan"))
    _lib.check(lib.hn_conv2d_dgrad_planes(P(gb), gb.numel(), None, P(wt3), P(dx1), P(skip), cin, ctypes.byref(d), tile, s), "dgrad")
    assert _same_conv(dx1, dx0, k * k)
    if cin % 64 == 0 and cout % 64 == 0 and tile < 2:
        dw0, dw1 = torch.empty_like(wt), torch.full_like(wt, float("nan"))
        _lib.check(lib.hn_conv2d_wgrad(P(x), P(gy), P(dw0), ctypes.byref(d), 1, vb, 0, s), "wgrad")
        _lib.check(lib.hn_conv2d_wgrad_planes(P(xb), xb.numel(), None, P(gb), gb.numel(), None, P(dw1), ctypes.byref(d), 1, 2 * tile, 0, s),
                   "wgrad_planes")
        assert torch.equal(dw0, dw1)


@pytest.mark.parametrize("math", ["f16x2", "bf16"])
def test_bn_backward_reduction_in_dgrad_epilogue(dev, math):
    """The first pass of a BatchNorm layer's backward (sum du, sum du * xhat, max |du|) taken by the data-gradient epilogue of
    the convolution that consumes the layer's output (hn_conv2d_dgrad_planes_bnreduce, ops._BnLink) against the stand-alone
    reduction pass (HEATNET_BN_REDUCE_FUSION off): a layer of three bottlenecks (conv1 / conv2 without, conv3 with a residual
    and byte ReLU masks, a downsample branch in the first), every parameter gradient and the input gradient.  The two paths sum
    the same du in a different order: fp32-noise agreement.  Then a tensor with TWO gradient contributions: the sums the
    epilogue took of the partial gradient must be discarded."""
    from heatnet_amd import ops, resnet
    torch.manual_seed(11)

    def run(fuse, two_consumers=False):
        prev, ops.BN_REDUCE_FUSION = ops.BN_REDUCE_FUSION, fuse
        for k_ in ops.bn_reduce_counts:
            ops.bn_reduce_counts[k_] = 0
        try:
            net = resnet.ResNet(resnet.Bottleneck, (3, 1, 1, 1), dual=False)
            layer = net.layer1.to(dev).train()
            g = torch.Generator().manual_seed(3)
            with torch.no_grad():
                for p_ in layer.parameters():
                    if p_.dim() == 4:
                        p_.copy_((torch.randn(p_.shape, generator=g) / np.sqrt(p_[0].numel())).to(dev))
                    else:
                        p_.copy_((torch.rand(p_.shape, generator=g) + 0.5).to(dev))
            x = F.relu(torch.randn(2, 64, 40, 48, generator=g)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            z = resnet.run_layer(layer, x * 1.0)
            gz = (torch.randn(z.shape, generator=g) * 1e-2).to(dev).contiguous(memory_format=torch.channels_last)
            if two_consumers:
                # the first block's bn1 output feeds conv2 -- and, here, a second reader: its dz is a SUM autograd builds
                blk = layer[0]
                y1, xa = resnet._first_conv(blk, x * 1.0)
                y2 = resnet.conv_bn_act(y1, blk.conv2, blk.bn2, ops.ACT_RELU, out="planes")
                g64 = gz[:, :64]
                ((ops.as_f32(y2) * g64).sum() + (ops.as_f32(y1) * g64).sum() + (ops.as_f32(xa) * 0).sum()).backward()
            else:
                z.backward(gz)
            grads = {n_: p_.grad.detach().float().cpu() for n_, p_ in layer.named_parameters() if p_.grad is not None}
            grads["x"] = x.grad.detach().float().cpu()
            return grads, dict(ops.bn_reduce_counts)
        finally:
            ops.BN_REDUCE_FUSION = prev
    prev_math = ops.set_conv_math(math)
    try:
        a, ca = run(True)
        b, cb = run(False)
        a2, ca2 = run(True, two_consumers=True)
        b2, cb2 = run(False, two_consumers=True)
    finally:
        ops.set_conv_math(prev_math)
    assert cb["fused"] == 0 and ca["fused"] >= 7, (ca, cb)  # 3 x (bn1, bn2) + bn3 of the first two blocks (+ nothing behind the last)
    assert ca["discarded"] == 0
    # (bf16: dy is STORED in bf16 -- a last-bit difference of the channel means moves roundings of dy, 2^-9 each)
    tol = 1e-2 if math == "bf16" else 2e-5
    for tag, got, ref in (("", a, b), ("two consumers", a2, b2)):
        for key in ref:
            err = float((got[key] - ref[key]).abs().max())
            assert err <= tol * max(float(ref[key].abs().max()), 1e-30), (tag, key, err, float(ref[key].abs().max()))
    assert ca2["discarded"] >= 1, ca2
