"""The streaming fp32-MFMA kernels for layers with a tiny channel count on one side (csrc/conv_small.hip): the 7x7 / stride-2
stems (cm/models/extractors.py:108-126) and the 13-class head (cm/models/pspnet.py:56-58), against fp64 torch convolutions on
the CPU and against the generic implicit-GEMM kernels they replace."""
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


def rel(a, ref):
    return float((a.double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("cin", [3, 1])
@pytest.mark.parametrize("hw", [(64, 96), (37, 53), (320, 704)])
def test_stem_forward_and_weight_gradient(dev, cin, hw):
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    s = _stream()
    h, w_ = hw
    n = 2
    ho, wo = (h - 1) // 2 + 1, (w_ - 1) // 2 + 1
    x = rnd(n, cin, h, w_, seed=1)
    wt = rnd(64, cin, 7, 7, seed=2, scale=1.0 / np.sqrt(49 * cin))
    gy = rnd(n, 64, ho, wo, seed=3) * 1e-2
    xr, wr = x.double(), wt.double().requires_grad_()
    yr = F.conv2d(xr, wr, None, stride=2, padding=3)
    yr.backward(gy.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = wt.permute(0, 2, 3, 1).contiguous().to(dev)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    d = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, 64, 64, 7, 7, 2, 3, 1)
    y = torch.full((n, ho, wo, 64), float("nan"), device=dev)
    stats = torch.zeros(ops.STAT_REPLICAS * 2 * 64, device=dev, dtype=torch.float64)
    _lib.check(lib.hn_conv_stem_fwd(P(xd), P(wd), P(y), P(stats), ctypes.byref(d), s), "stem fwd")
    assert rel(y.permute(0, 3, 1, 2), yr.detach()) < 2e-6
    st = stats.view(ops.STAT_REPLICAS, 2, 64).sum(0).cpu()
    yd = yr.detach()
    assert float((st[0] - yd.sum((0, 2, 3))).abs().max()) <= 1e-5 * float(yd.abs().sum((0, 2, 3)).max())
    assert float((st[1] - (yd * yd).sum((0, 2, 3))).abs().max()) <= 1e-5 * float((yd * yd).sum((0, 2, 3)).max())
    dw = torch.full((64, 7, 7, cin), float("nan"), device=dev)
    _lib.check(lib.hn_conv_stem_wgrad(P(xd), P(gd), P(dw), ctypes.byref(d), 0, s), "stem wgrad")
    assert rel(dw.permute(0, 3, 1, 2), wr.grad) < 5e-6
    _lib.check(lib.hn_conv_stem_wgrad(P(xd), P(gd), P(dw), ctypes.byref(d), 2, s), "stem wgrad accumulate")  # HN_WGRAD_ACCUMULATE
    assert rel(dw.permute(0, 3, 1, 2), 2 * wr.grad) < 5e-6
    # refusals
    d5 = _lib.HnConvDesc(n, h, w_, cin, cin, ho, wo, 64, 64, 5, 5, 2, 3, 1)
    assert lib.hn_conv_stem_fwd(P(xd), P(wd), P(y), None, ctypes.byref(d5), s) == -4


@pytest.mark.parametrize("cout", [13, 16, 5])
@pytest.mark.parametrize("hw", [(40, 44), (33, 47), (320, 704)])
def test_narrow_head_gradients(dev, cout, hw):
    from heatnet_amd import _lib
    lib = _lib.load()
    s = _stream()
    h, w_ = hw
    n = 2
    x = rnd(n, 64, h, w_, seed=4)
    wt = rnd(cout, 64, 1, 1, seed=5, scale=0.125)
    gy = rnd(n, cout, h, w_, seed=6) * 1e-3
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    F.conv2d(xr, wr).backward(gy.double())
    xd = x.permute(0, 2, 3, 1).contiguous().to(dev)
    wd = wt.reshape(cout, 64).contiguous().to(dev)
    gd = gy.permute(0, 2, 3, 1).contiguous().to(dev)
    d = _lib.HnConvDesc(n, h, w_, 64, 64, h, w_, cout, cout, 1, 1, 1, 0, 1)
    dx = torch.full((n, h, w_, 64), float("nan"), device=dev)
    _lib.check(lib.hn_conv_narrow_dgrad(P(gd), P(wd), P(dx), ctypes.byref(d), s), "narrow dgrad")
    assert rel(dx.permute(0, 3, 1, 2), xr.grad) < 2e-6
    dw = torch.full((cout, 64), float("nan"), device=dev)
    _lib.check(lib.hn_conv_narrow_wgrad(P(xd), P(gd), P(dw), ctypes.byref(d), 0, s), "narrow wgrad")
    assert rel(dw, wr.grad[:, :, 0, 0]) < 5e-6
    dw.zero_()
    _lib.check(lib.hn_conv_narrow_wgrad(P(xd), P(gd), P(dw), ctypes.byref(d), 1, s), "narrow wgrad (dw zeroed)")
    assert rel(dw, wr.grad[:, :, 0, 0]) < 5e-6
    d32 = _lib.HnConvDesc(n, h, w_, 32, 32, h, w_, cout, cout, 1, 1, 1, 0, 1)
    assert lib.hn_conv_narrow_dgrad(P(gd), P(wd), P(dx), ctypes.byref(d32), s) == -4


def test_small_kernels_through_autograd_match_the_generic_kernels(dev):
    """ops.conv2d on a stem (with the BatchNorm statistics epilogue) and on the head (bias, data and weight gradient) with the
    streaming kernels on and off."""
    from heatnet_amd import ops
    outs = {}
    for on in (True, False):
        prev, prev_stem = ops.SMALL_KERNELS, ops.SMALL_STEM
        ops.SMALL_KERNELS = ops.SMALL_STEM = on
        try:
            x = rnd(2, 3, 72, 104, seed=7).to(dev)
            w = rnd(64, 3, 7, 7, seed=8, scale=0.1).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            stats = torch.zeros(ops.STAT_REPLICAS * 2 * 64, device=dev, dtype=torch.float64)
            y = ops.conv2d(x, w, None, 2, 3, 1, want_stats=stats)
            y.backward(rnd(*y.shape, seed=9).to(dev))
            xh = rnd(2, 64, 72, 104, seed=10).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            wh = rnd(13, 64, 1, 1, seed=11, scale=0.1).to(dev).requires_grad_()
            bh = rnd(13, seed=12).to(dev).requires_grad_()
            yh = ops.conv2d(xh, wh, bh, 1, 0, 1)
            yh.backward(rnd(*yh.shape, seed=13).to(dev))
            outs[on] = [t.detach().float().cpu() for t in (y, stats.view(ops.STAT_REPLICAS, 2, 64).sum(0), w.grad, yh, xh.grad, wh.grad,
                                                          bh.grad)]
        finally:
            ops.SMALL_KERNELS, ops.SMALL_STEM = prev, prev_stem
    for a, b in zip(outs[True], outs[False]):
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max())
