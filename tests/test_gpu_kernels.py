"""GPU: every HIP kernel (through the C ABI / autograd wrappers) against the same op in stock torch CPU fp32.

Tolerances are written per test; convolution sums up to K = 9216 products in a different order than the CPU,
so its bound scales with sqrt(K) * eps_fp32 * |x||w|.
"""
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


def _ops():
    from heatnet_amd import ops
    return ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def close(a, b, atol, rtol=0.0, what=""):
    a = a.detach().cpu().float()
    b = b.detach().cpu().float()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs()
    bound = atol + rtol * b.abs()
    bad = err > bound
    assert not bad.any(), "%s: max err %.3e (bound %.3e) at %d/%d elements" % (
        what, err.max().item(), bound.min().item(), int(bad.sum()), bad.numel())


CONV_CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil, bias
    (2, 64, 20, 24, 256, 1, 1, 0, 1, False),     # 1x1, vector path
    (1, 256, 9, 13, 64, 1, 1, 0, 1, False),      # M tail (117 pixels)
    (2, 64, 16, 20, 64, 3, 1, 1, 1, True),       # 3x3 + bias
    (1, 128, 12, 20, 128, 3, 1, 2, 2, False),    # dilation 2
    (1, 64, 17, 23, 96, 3, 1, 4, 4, False),      # dilation 4, odd sizes, Cout tail (96)
    (2, 128, 16, 24, 128, 3, 2, 1, 1, False),    # stride 2
    (2, 256, 16, 24, 512, 1, 2, 0, 1, False),    # 1x1 stride 2 (downsample)
    (2, 3, 32, 48, 64, 7, 2, 3, 1, False),       # RGB stem, scalar path
    (2, 1, 32, 48, 64, 7, 2, 3, 1, False),       # IR stem
    (2, 13, 32, 32, 64, 4, 2, 1, 1, True),       # critic conv1 on logits
    (2, 64, 16, 16, 128, 4, 2, 1, 1, True),      # critic conv2
    (2, 512, 4, 6, 1, 4, 2, 1, 1, True),         # critic classifier (Cout = 1)
    (2, 136, 12, 16, 72, 3, 1, 1, 1, True),      # wide ragged Cin (ResNeXt decoder: 525): zero-padded to 192 channels
    (2, 136, 12, 16, 136, 3, 1, 1, 1, True),     # ... and ragged Cout: run at 192 x 192, the output is the narrow view
    (2, 512, 64, 88, 64, 4, 2, 1, 1, True),      # critic conv1 on deep features: 2816 rows, 256 K steps -> split-K above 2048 rows
    (2, 64, 12, 16, 13, 1, 1, 0, 1, True),       # final 1x1 (Cout = 13)
    (1, 2048, 6, 6, 2048, 1, 1, 0, 1, False),    # PSP stage conv on a 6x6 map
]


@pytest.mark.parametrize("conv_math", ["f32", "bf16x3", "f16x2"], indirect=True)
@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_conv2d_forward_backward(dev, case, conv_math):
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil, bias = case
    x = rnd(n, cin, h, w, seed=1)
    wt = rnd(cout, cin, k, k, seed=2, scale=1.0 / np.sqrt(cin * k * k))
    b = rnd(cout, seed=3) if bias else None
    xr, wr = x.clone().requires_grad_(), wt.clone().requires_grad_()
    br = b.clone().requires_grad_() if bias else None
    yr = F.conv2d(xr, wr, br, stride=stride, padding=pad, dilation=dil)
    gy = rnd(*yr.shape, seed=4)
    yr.backward(gy)

    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    bd = b.to(dev).requires_grad_() if bias else None
    y, stats = ops.conv2d(xd, wd, bd, stride, pad, dil, want_stats=True)
    tol = 3e-6 * np.sqrt(cin * k * k) + 2e-6
    close(y, yr, atol=tol * 4, what="y")
    npix = yr.shape[0] * yr.shape[2] * yr.shape[3]
    stats = stats.view(ops.STAT_REPLICAS, 2, cout).sum(0)  # replicas of {sum, sumsq}
    close(stats[0], yr.sum((0, 2, 3)), atol=tol * 4 * np.sqrt(npix) + 1e-4, rtol=1e-5, what="stats.sum")
    close(stats[1], (yr * yr).sum((0, 2, 3)), atol=1e-4, rtol=1e-4, what="stats.sumsq")
    y.backward(gy.to(dev))
    ktot = cout * k * k
    close(xd.grad, xr.grad, atol=3e-6 * np.sqrt(ktot) * 4 + 2e-6, what="dx")
    close(wd.grad, wr.grad, atol=2e-5 * np.sqrt(npix) + 1e-5, rtol=1e-4, what="dw")
    if bias:
        close(bd.grad, br.grad, atol=1e-5 * np.sqrt(npix) + 1e-5, rtol=1e-5, what="db")


@pytest.mark.parametrize("case", [(2, 256, 24, 40, 256, 3, 1, 2, 2), (2, 1024, 20, 24, 256, 1, 1, 0, 1),
                                  (2, 128, 32, 48, 128, 3, 2, 1, 1), (2, 64, 40, 48, 256, 1, 1, 0, 1)], ids=str)
def test_conv2d_bf16x3_is_fp32_accurate(dev, case):
    """The bf16x3 kernels (exact 3-way bf16 split, 6 MFMAs, fp32 accumulate) and the f16x2 kernels (power-of-two tensor
    scales, 2-way fp16 split, 3 MFMAs) against an fp64 convolution: their error must not exceed the fp32-MFMA kernels'
    (all are dominated by fp32 accumulation rounding)."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = F.relu(rnd(n, cin, h, w, seed=31))                      # post-ReLU activations, as in the network
    wt = rnd(cout, cin, k, k, seed=32, scale=1.0 / np.sqrt(cin * k * k))
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    yr = F.conv2d(xr, wr, None, stride=stride, padding=pad, dilation=dil)
    gy = rnd(*yr.shape, seed=33)
    yr.backward(gy.double())
    errs = {}
    for math in ("f32", "bf16x3", "f16x2"):
        prev = ops.set_conv_math(math)
        try:
            xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            y = ops.conv2d(xd, wd, None, stride, pad, dil)
            y.backward(gy.to(dev))
            errs[math] = tuple(float((a.detach().cpu().double() - b.detach()).pow(2).mean().sqrt())
                               for a, b in ((y, yr), (xd.grad, xr.grad), (wd.grad, wr.grad)))
        finally:
            ops.set_conv_math(prev)
    for math in ("bf16x3", "f16x2"):
        for what, e32, ex in zip(("y", "dx", "dw"), errs["f32"], errs[math]):
            assert ex <= 1.25 * e32 + 1e-9, "%s: rms error vs fp64 %.3e (%s) > 1.25 x %.3e (fp32 MFMA)" % (what, ex, math, e32)


def test_conv2d_fused_leaky_and_channel_slice_input(dev):
    """LeakyReLU in the epilogue (critics) and an input that is a channel slice of a wider NHWC buffer (ld > C)."""
    ops = _ops()
    big = rnd(2, 96, 12, 16, seed=5)
    wt = rnd(32, 64, 4, 4, seed=6, scale=0.03)
    b = rnd(32, seed=7)
    xr = big[:, 16:80].clone().requires_grad_()
    wr, br = wt.clone().requires_grad_(), b.clone().requires_grad_()
    yr = F.leaky_relu(F.conv2d(xr, wr, br, stride=2, padding=1), 0.2)
    gy = rnd(*yr.shape, seed=8)
    yr.backward(gy)
    bigd = big.to(dev).contiguous(memory_format=torch.channels_last)
    xd = bigd[:, 16:80].detach().requires_grad_()
    assert ops.nhwc_ld(xd) == 96
    wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    bd = b.to(dev).requires_grad_()
    y = ops.conv2d(xd, wd, bd, 2, 1, 1, act=ops.ACT_LEAKY, slope=0.2)
    close(y, yr, atol=2e-4, what="y")
    y.backward(gy.to(dev))
    close(xd.grad, xr.grad, atol=2e-4, what="dx")
    close(wd.grad, wr.grad, atol=5e-4, rtol=1e-4, what="dw")
    close(bd.grad, br.grad, atol=5e-4, rtol=1e-4, what="db")


@pytest.mark.parametrize("act", ["none", "relu", "prelu"])
@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("residual", [False, True])
def test_bn_act(dev, act, training, residual):
    ops = _ops()
    if residual and act == "prelu":
        pytest.skip("the reference never adds a residual before PReLU")
    n, c, h, w = 3, 64, 10, 14
    y = rnd(n, c, h, w, seed=1, scale=2.0) + 0.5
    gamma, beta = rnd(c, seed=2).abs() + 0.5, rnd(c, seed=3) * 0.1
    rm, rv = rnd(c, seed=4) * 0.1, rnd(c, seed=5).abs() + 0.5
    res = rnd(n, c, h, w, seed=6) if residual else None
    slope = torch.tensor([0.25])

    yr, gr, br = y.clone().requires_grad_(), gamma.clone().requires_grad_(), beta.clone().requires_grad_()
    rr = res.clone().requires_grad_() if residual else None
    sr = slope.clone().requires_grad_()
    rm_r, rv_r = rm.clone(), rv.clone()
    u = F.batch_norm(yr, rm_r, rv_r, gr, br, training=training, momentum=0.1, eps=1e-5)
    if residual:
        u = u + rr
    zr = {"none": u, "relu": F.relu(u), "prelu": F.prelu(u, sr)}[act]
    gz = rnd(n, c, h, w, seed=7)
    zr.backward(gz)

    cl = torch.channels_last
    yd = y.to(dev).contiguous(memory_format=cl).requires_grad_()
    gd, bd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
    rd = res.to(dev).contiguous(memory_format=cl).requires_grad_() if residual else None
    sd = slope.to(dev).requires_grad_()
    rm_d, rv_d = rm.to(dev), rv.to(dev)
    code = {"none": ops.ACT_NONE, "relu": ops.ACT_RELU, "prelu": ops.ACT_PRELU}[act]
    z = ops.bn_act(yd, None, gd, bd, rm_d, rv_d, residual=rd, prelu_weight=sd if act == "prelu" else None,
                   training=training, act=code)
    close(z, zr, atol=2e-5, rtol=1e-5, what="z")
    close(rm_d, rm_r, atol=1e-6, rtol=1e-5, what="running_mean")
    close(rv_d, rv_r, atol=1e-5, rtol=1e-5, what="running_var")
    z.backward(gz.to(dev))
    close(yd.grad, yr.grad, atol=3e-5, rtol=1e-4, what="dy")
    close(gd.grad, gr.grad, atol=2e-3, rtol=1e-4, what="dgamma")
    close(bd.grad, br.grad, atol=2e-3, rtol=1e-4, what="dbeta")
    if residual:
        close(rd.grad, rr.grad, atol=1e-6, what="dres")
    if act == "prelu":
        close(sd.grad, sr.grad, atol=2e-3, rtol=1e-4, what="dslope")


@pytest.mark.parametrize("act", ["relu", "prelu"])
def test_bn_act_with_folded_dropout_mask(dev, act):
    """Dropout2d folded into the BN/activation pass == BatchNorm -> act -> per-(image, channel) scaling
    (reference pspnet.py:66-72: drop_2 after each PSPUpsample)."""
    ops = _ops()
    n, c, h, w = 3, 64, 9, 11
    y = rnd(n, c, h, w, seed=11, scale=2.0) + 0.3
    gamma, beta = rnd(c, seed=12).abs() + 0.5, rnd(c, seed=13) * 0.1
    mask = (torch.from_numpy(np.random.RandomState(5).uniform(size=(n, c))) > 0.3).float() / 0.7
    slope = torch.tensor([0.25])
    gz = rnd(n, c, h, w, seed=14)

    yr, gr, br, sr = (t.clone().requires_grad_() for t in (y, gamma, beta, slope))
    u = F.batch_norm(yr, torch.zeros(c), torch.ones(c), gr, br, training=True, momentum=0.1, eps=1e-5)
    zr = (F.relu(u) if act == "relu" else F.prelu(u, sr)) * mask[:, :, None, None]
    zr.backward(gz)

    yd = y.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    gd, bd, sd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_(), slope.to(dev).requires_grad_()
    z = ops.bn_act(yd, None, gd, bd, torch.zeros(c, device=dev), torch.ones(c, device=dev),
                   prelu_weight=sd if act == "prelu" else None, training=True,
                   act=ops.ACT_RELU if act == "relu" else ops.ACT_PRELU, drop_mask=mask.to(dev))
    close(z, zr, atol=2e-5, rtol=1e-5, what="z")
    z.backward(gz.to(dev))
    close(yd.grad, yr.grad, atol=3e-5, rtol=1e-4, what="dy")
    close(gd.grad, gr.grad, atol=2e-3, rtol=1e-4, what="dgamma")
    close(bd.grad, br.grad, atol=2e-3, rtol=1e-4, what="dbeta")
    if act == "prelu":
        close(sd.grad, sr.grad, atol=2e-3, rtol=1e-4, what="dslope")


@pytest.mark.parametrize("shape", [(2, 64, 10, 14), (1, 12, 7, 5), (2, 6, 40, 88)])
def test_pyramid_pool_fused_backward(dev, shape):
    """All pyramid levels' adaptive poolings (reference pspnet.py:22-24) with their gradients scattered back in one pass;
    7x5 and 40x88 have overlapping bins at sizes 2, 3, 6."""
    ops = _ops()
    n, c, h, w = shape
    sizes = (1, 2, 3, 6)
    x = rnd(n, c, h, w, seed=21)
    if h >= 6 and w >= 6:  # the one-read forward (hn_pyramid_pool_fwd; off by default) against the level-by-level kernels
        prev = ops.PYRAMID_FWD_FUSED
        try:
            xg = x.to(dev).contiguous(memory_format=torch.channels_last)
            ops.PYRAMID_FWD_FUSED = True
            ya = ops.pyramid_pool(xg, sizes)
            ops.PYRAMID_FWD_FUSED = False
            yb = ops.pyramid_pool(xg, sizes)
        finally:
            ops.PYRAMID_FWD_FUSED = prev
        for a, b, s in zip(ya, yb, sizes):
            close(a, b.cpu(), atol=1e-6, rtol=1e-5, what="one-read pool%d" % s)
    gs = [rnd(n, c, s, s, seed=22 + s) for s in sizes]
    xr = x.clone().requires_grad_()
    yr = [F.adaptive_avg_pool2d(xr, (s, s)) for s in sizes]
    sum((y * g).sum() for y, g in zip(yr, gs)).backward()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    yd = ops.pyramid_pool(xd, sizes)
    for a, b, s in zip(yd, yr, sizes):
        close(a, b, atol=1e-6, rtol=1e-5, what="pool%d" % s)
    sum((y * g.to(dev)).sum() for y, g in zip(yd, gs)).backward()
    close(xd.grad, xr.grad, atol=1e-6, rtol=1e-5, what="dx")
    # a level without gradient is skipped
    xd2 = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y2 = ops.pyramid_pool(xd2, sizes)
    (y2[1] * gs[1].to(dev)).sum().backward()
    xr.grad = None
    (F.adaptive_avg_pool2d(xr, (2, 2)) * gs[1]).sum().backward()
    close(xd2.grad, xr.grad, atol=1e-6, rtol=1e-5, what="dx(level 2 only)")


@pytest.mark.parametrize("shape", [(2, 64, 40, 88), (1, 12, 7, 9), (3, 8, 24, 40), (2, 64, 6, 6)])
def test_pyramid_merge_forward_and_fused_backward(dev, shape):
    """ops.pyramid_merge = relu(main + sum_k bilinear_up(prior_k)) (reference pspnet.py:22-25 after the bottleneck was applied per
    level) against stock torch on the CPU; the backward through hn_pyramid_merge_bwd (one read of the gradient: du and every
    prior's gradient from row sums, then columns of row sums) and, with the knob off, level by level through hn_bilinear_bwd."""
    ops = _ops()
    n, c, h, w = shape
    sizes = (1, 2, 3, 6)
    main = rnd(n, c, h, w, seed=31)
    priors = [rnd(n, c, s, s, seed=32 + s) for s in sizes]
    gy = rnd(n, c, h, w, seed=39)
    mr = main.double().requires_grad_()
    pr = [p.double().requires_grad_() for p in priors]
    ref = F.relu(mr + sum(F.interpolate(p, size=(h, w), mode="bilinear", align_corners=False) for p in pr))
    ref.backward(gy.double())
    prev = ops.PYRAMID_BWD_FUSED
    try:
        for fused in (True, False):
            ops.PYRAMID_BWD_FUSED = fused
            md = main.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            pd = [p.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_() for p in priors]
            out = ops.pyramid_merge(md, pd)
            close(out, ref.detach().float(), atol=2e-6, rtol=1e-5, what="out")
            out.backward(gy.to(dev))
            close(md.grad, mr.grad.float(), atol=1e-6, rtol=1e-5, what="dmain fused=%s" % fused)
            for a, b, s in zip(pd, pr, sizes):
                close(a.grad, b.grad.float(), atol=2e-5 * float(b.grad.abs().max()), rtol=1e-5, what="dprior%d fused=%s" % (s, fused))
    finally:
        ops.PYRAMID_BWD_FUSED = prev


def test_bn_accumulators_recover_from_an_interrupted_step(dev):
    """The per-layer double accumulators must be zero when a kernel starts adding into them.  If an exception interrupts a
    step between the producer (conv epilogue / backward reduction) and the consumer that re-zeroes them, the host-side
    guards of ops.BnWorkspace clear the leftovers at the next use instead of silently folding them into the statistics."""
    ops = _ops()
    from heatnet_amd.layers import HnBatchNorm2d, HnConv2d, conv_bn_act
    torch.manual_seed(1)
    conv, bn = HnConv2d(32, 64, 3, padding=1, bias=False).to(dev), HnBatchNorm2d(64).to(dev)
    x = torch.randn(2, 32, 10, 12, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    g = torch.randn(2, 64, 10, 12, device=dev).contiguous(memory_format=torch.channels_last)
    z0 = conv_bn_act(x, conv, bn, ops.ACT_RELU)
    z0.backward(g)
    gx0, gg0 = x.grad.clone(), bn.weight.grad.clone()
    ws = bn.workspace(dev)
    ws.open_stats()          # an interrupted forward: producer ran, consumer did not
    ws.stats.fill_(123.0)
    ws.open_bwd()            # an interrupted backward
    ws.bwd.fill_(-7.0)
    x.grad = None
    bn.weight.grad = None
    z1 = conv_bn_act(x, conv, bn, ops.ACT_RELU)
    z1.backward(g)
    close(z1, z0, atol=1e-6, what="z after an interrupted forward")
    close(x.grad, gx0, atol=1e-6, what="dx after an interrupted backward")
    close(bn.weight.grad, gg0, atol=1e-5, rtol=1e-5, what="dgamma after an interrupted backward")


def test_bn_stats_from_conv_epilogue_match_standalone(dev):
    """conv_bn_act (statistics from the conv epilogue) == conv then BatchNorm on CPU, incl. narrow C with ld tails."""
    from heatnet_amd.layers import HnBatchNorm2d, HnConv2d, conv_bn_act
    ops = _ops()
    conv = HnConv2d(64, 128, 3, padding=1, bias=False)
    bn = HnBatchNorm2d(128)
    ref_conv = torch.nn.Conv2d(64, 128, 3, padding=1, bias=False)
    ref_bn = torch.nn.BatchNorm2d(128)
    ref_conv.load_state_dict(conv.state_dict())
    bn.weight.data = rnd(128, seed=1).abs() + 0.5
    ref_bn.load_state_dict(bn.state_dict())
    x = rnd(2, 64, 12, 20, seed=2)
    zr = F.relu(ref_bn(ref_conv(x)))
    conv, bn = conv.to(dev), bn.to(dev)
    z = conv_bn_act(x.to(dev), conv, bn, ops.ACT_RELU)
    close(z, zr, atol=5e-5, rtol=1e-4, what="z")
    sd = bn.state_dict()
    close(sd["running_var"], ref_bn.running_var, atol=1e-5, rtol=1e-4, what="running_var")
    assert int(sd["num_batches_tracked"]) == 1


@pytest.mark.parametrize("shape", [(2, 64, 20, 28), (1, 64, 21, 27), (2, 3, 9, 11)])
def test_maxpool(dev, shape):
    ops = _ops()
    x = rnd(*shape, seed=1)
    xr = x.clone().requires_grad_()
    yr = F.max_pool2d(xr, 3, 2, 1)
    g = rnd(*yr.shape, seed=2)
    yr.backward(g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = ops.maxpool3x3s2(xd)
    close(y, yr, atol=0, what="y")
    y.backward(g.to(dev))
    close(xd.grad, xr.grad, atol=1e-6, what="dx")


@pytest.mark.parametrize("size", [1, 2, 3, 6])
@pytest.mark.parametrize("hw", [(40, 88), (8, 12), (5, 7)])
def test_adaptive_avgpool(dev, size, hw):
    ops = _ops()
    x = rnd(2, 64, *hw, seed=1)
    xr = x.clone().requires_grad_()
    yr = F.adaptive_avg_pool2d(xr, (size, size))
    g = rnd(*yr.shape, seed=2)
    yr.backward(g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = ops.adaptive_avg_pool(xd, size)
    close(y, yr, atol=2e-6, rtol=1e-5, what="y")
    y.backward(g.to(dev))
    close(xd.grad, xr.grad, atol=1e-6, rtol=1e-5, what="dx")


BILINEAR_CASES = [((2, 64, 10, 14), (20, 28)), ((1, 12, 5, 7), (10, 14)), ((2, 3, 1, 4), (2, 8)), ((1, 8, 9, 1), (18, 2)), ((1, 64, 1, 1), (40, 88)), ((2, 64, 2, 2), (40, 88)),
                  ((2, 64, 3, 3), (40, 88)), ((2, 64, 6, 6), (40, 88)), ((2, 64, 6, 6), (8, 12)),
                  ((2, 1, 2, 5), (64, 160)), ((2, 1, 1, 2), (32, 64)), ((1, 13, 7, 9), (14, 18))]


@pytest.mark.parametrize("case", BILINEAR_CASES, ids=[str(c) for c in BILINEAR_CASES])
def test_bilinear(dev, case):
    ops = _ops()
    shape, size = case
    x = rnd(*shape, seed=1)
    xr = x.clone().requires_grad_()
    yr = F.interpolate(xr, size=size, mode="bilinear", align_corners=False)
    g = rnd(*yr.shape, seed=2)
    yr.backward(g)
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = ops.bilinear(xd, size)
    close(y, yr, atol=2e-6, rtol=1e-5, what="y")
    y.backward(g.to(dev))
    scale = (size[0] * size[1]) / (shape[2] * shape[3])
    close(xd.grad, xr.grad, atol=2e-6 * max(1.0, scale), rtol=1e-4, what="dx")


def test_upsample_scale_factor_32_matches_nn_upsample(dev):
    ops = _ops()
    x = rnd(2, 1, 10, 22, seed=3)
    yr = torch.nn.Upsample(scale_factor=32, mode="bilinear")(x)
    y = ops.bilinear(x.to(dev), (320, 704))
    close(y, yr, atol=2e-6, what="x32")


@pytest.mark.parametrize("ignore", [-100, 13])
def test_cross_entropy(dev, ignore):
    ops = _ops()
    n, c, h, w = 2, 13, 24, 40
    logits = rnd(n, c, h, w, seed=1, scale=3.0)
    g = torch.Generator().manual_seed(2)
    labels = torch.randint(0, 14 if ignore == 13 else 13, (n, h, w), generator=g)
    lr = logits.clone().requires_grad_()
    loss_r = F.cross_entropy(lr, labels, ignore_index=ignore)
    (loss_r * 1.7).backward()
    ld = logits.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    loss = ops.cross_entropy(ld, labels.to(dev), ignore)
    assert abs(loss.item() - loss_r.item()) < 2e-6 * max(1.0, abs(loss_r.item())) * 4
    (loss * 1.7).backward()
    close(ld.grad, lr.grad, atol=1e-8, rtol=2e-5, what="dlogits")


@pytest.mark.parametrize("kind", ["MSE", "BCE"])
@pytest.mark.parametrize("target", [0.0, 1.0])
def test_adv_loss(dev, kind, target):
    ops = _ops()
    x = rnd(2, 1, 64, 96, seed=1, scale=2.0)
    xr = x.clone().requires_grad_()
    crit = torch.nn.BCEWithLogitsLoss() if kind == "BCE" else torch.nn.MSELoss()
    lr = crit(xr, torch.full_like(xr, target))
    (lr * 0.3).backward()
    xd = x.to(dev).requires_grad_()
    l = ops.adv_loss(xd, target, kind)
    assert abs(l.item() - lr.item()) < 1e-5 * max(1.0, abs(lr.item()))
    (l * 0.3).backward()
    close(xd.grad, xr.grad, atol=1e-9, rtol=2e-5, what="dx")


def test_concat_dropout_mask_and_add(dev):
    ops = _ops()
    a, b = rnd(2, 64, 6, 10, seed=1), rnd(2, 32, 6, 10, seed=2)
    ar, br = a.clone().requires_grad_(), b.clone().requires_grad_()
    mask = (torch.rand(2, 96, generator=torch.Generator().manual_seed(3)) > 0.3).float() / 0.7
    yr = torch.cat([ar, br], 1) * mask[:, :, None, None]
    g = rnd(*yr.shape, seed=4)
    yr.backward(g)
    cl = torch.channels_last
    ad = a.to(dev).contiguous(memory_format=cl).requires_grad_()
    bd = b.to(dev).contiguous(memory_format=cl).requires_grad_()
    y = ops.dropout2d(ops.concat([ad, bd]), 0.3, True, mask=mask.to(dev))
    close(y, yr, atol=1e-7, what="y")
    y.backward(g.to(dev))
    close(ad.grad, ar.grad, atol=1e-7, what="da")
    close(bd.grad, br.grad, atol=1e-7, what="db")


def test_argmax_confusion_bit_exact(dev):
    from heatnet_amd import miou
    from oracle import heatnet_ref as R
    logits = rnd(2, 13, 40, 56, seed=1)
    labels = torch.randint(0, 14, (2, 40, 56), generator=torch.Generator().manual_seed(2))
    ops = _ops()
    pred, conf = ops.argmax_confusion(logits.to(dev), labels.to(dev), 14)
    ref_pred = logits.argmax(1)
    assert torch.equal(pred.cpu().long(), ref_pred)
    ref_conf = np.bincount((ref_pred + 14 * labels).view(-1).numpy(), minlength=196).reshape(14, 14)
    np.testing.assert_array_equal(conf.cpu().numpy(), ref_conf)
    np.testing.assert_array_equal(miou.ious_from_confusion(conf.cpu().numpy()),
                                  R.calculate_ious(ref_pred.numpy(), labels.numpy()))
    iou, m = miou.iou_eval_value(conf.cpu().numpy(), (12, 13))
    riou, rm = R.confusion_matrix_iou(ref_pred.numpy(), labels.numpy(), 14, (12, 13))
    np.testing.assert_array_equal(iou, riou)
    assert m == rm


def test_optimizer_steps(dev):
    ops = _ops()
    p, g = rnd(10007, seed=1), rnd(10007, seed=2)
    pr = p.clone().requires_grad_()
    pr.grad = g.clone()
    opt = torch.optim.RMSprop([pr], lr=1e-4)
    pd, sq = p.to(dev), torch.zeros(10007, device=dev)
    for _ in range(3):
        opt.step()
        ops.rmsprop_step_(pd, g.to(dev), sq, 1e-4)
    close(pd, pr, atol=1e-7, rtol=1e-6, what="rmsprop")
    pr2 = p.clone().requires_grad_()
    pr2.grad = g.clone()
    opt2 = torch.optim.Adam([pr2], lr=1e-3)
    pd2, m, v = p.to(dev), torch.zeros(10007, device=dev), torch.zeros(10007, device=dev)
    for step in range(1, 4):
        opt2.step()
        ops.adam_step_(pd2, g.to(dev), m, v, step, 1e-3)
    close(pd2, pr2, atol=1e-6, rtol=1e-5, what="adam")


def test_c_abi_rejects_bad_arguments(dev):
    """Error behaviour of the C ABI: negative HN_E* codes, nothing thrown, nothing launched."""
    import ctypes
    from heatnet_amd import _lib
    lib = _lib.load()
    x = torch.zeros(1, 8, 8, 64, device=dev)
    w = torch.zeros(64, 1, 1, 64, device=dev)
    y = torch.zeros(1, 8, 8, 64, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    good = _lib.HnConvDesc(1, 8, 8, 64, 64, 8, 8, 64, 64, 1, 1, 1, 0, 1)
    assert lib.hn_conv2d_fwd(P(x), P(w), None, None, P(y), None, ctypes.byref(good), 0, 0.0, 0, s) == 0
    assert lib.hn_conv2d_fwd(None, P(w), None, None, P(y), None, ctypes.byref(good), 0, 0.0, 0, s) == -1      # HN_EINVAL
    assert lib.hn_conv2d_fwd(P(x), P(w), None, None, P(y), None, ctypes.byref(good), 7, 0.0, 0, s) == -1      # bad activation
    assert lib.hn_conv2d_fwd(P(x), P(w), None, None, P(y), None, ctypes.byref(good), 0, 0.0, 99, s) == -1     # bad variant
    bad_shape = _lib.HnConvDesc(1, 8, 8, 64, 64, 7, 8, 64, 64, 1, 1, 1, 0, 1)                           # Hout inconsistent
    assert lib.hn_conv2d_fwd(P(x), P(w), None, None, P(y), None, ctypes.byref(bad_shape), 0, 0.0, 0, s) == -4  # HN_ESHAPE
    bad_ld = _lib.HnConvDesc(1, 8, 8, 64, 32, 8, 8, 64, 64, 1, 1, 1, 0, 1)                              # ldx < Cin
    assert lib.hn_conv2d_fwd(P(x), P(w), None, None, P(y), None, ctypes.byref(bad_ld), 0, 0.0, 0, s) == -1
    assert lib.hn_bn_act_fwd(P(x), 64, None, None, None, 0, P(y), 64, 64, 64, 1, None, 0.0, None, 0, None, None, s) == -1
    assert lib.hn_bn_act_fwd(P(x), 64, P(w), P(w), None, 0, P(y), 64, 64, 64, 1, None, 0.0, P(w), 48, None, None, s) == -1  # hw !| npix
    assert lib.hn_pyramid_pool_bwd(9, None, None, None, None, P(y), 64, 1, 8, 8, 64, s) == -1            # > 8 levels
    assert lib.hn_ce_fwd(P(x), 64, P(y), 64, 64, -100, P(y), s) == -1                                   # C > 32 classes
    with pytest.raises(RuntimeError, match="invalid argument"):
        _lib.check(-1, "probe")
    torch.cuda.synchronize()


# ---- second bf16x3 schedule (conv_igemm2.hip), plain bf16 arithmetic, A-operand transform ---------------------------------

def _abi_conv(dev, x, wt, stride, pad, dil, variant, bias=None, act=0, slope=0.0, want_stats=False, affine=None):
    """hn_conv2d_fwd / hn_conv2d_fwd_affine through the C ABI with an explicit kernel variant; x NCHW cpu, wt OIHW cpu."""
    import ctypes
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    n, cin, h, w = x.shape
    cout, _, k, _ = wt.shape
    ho, wo = ops.conv_out_size(h, k, stride, pad, dil), ops.conv_out_size(w, k, stride, pad, dil)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous()
    wd = wt.to(dev).permute(0, 2, 3, 1).contiguous()
    nw = wd.numel()
    wx3 = torch.empty(3 * ops.x3_plane(nw), device=dev, dtype=torch.int16)
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    _lib.check(lib.hn_split_bf16x3(P(wd), P(wx3), nw, s), "split")
    y = torch.empty(n, ho, wo, cout, device=dev)
    stats = torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.float64) if want_stats else None
    bd = bias.to(dev) if bias is not None else None
    d = _lib.HnConvDesc(n, h, w, cin, cin, ho, wo, cout, cout, k, k, stride, pad, dil)
    if affine is None:
        _lib.check(lib.hn_conv2d_fwd(P(xd), P(wd), P(wx3), P(bd), P(y), P(stats), ctypes.byref(d), act, slope, variant, s),
                   "hn_conv2d_fwd")
    else:
        sc, sh = affine[0].to(dev), affine[1].to(dev)
        _lib.check(lib.hn_conv2d_fwd_affine(P(xd), P(sc), P(sh), P(wd), P(wx3), P(bd), P(y), P(stats), ctypes.byref(d),
                                            variant, s), "hn_conv2d_fwd_affine")
    torch.cuda.synchronize()
    return y.permute(0, 3, 1, 2), stats


X3_SCHEDULE_CASES = [
    # n, cin, h, w, cout, k, stride, pad, dil     (all with H*W >= 128 and Cin % 32 == 0: the bf16 kernels' domain)
    (2, 64, 40, 44, 256, 1, 1, 0, 1),      # 1x1, several 128x128 tiles
    (2, 256, 24, 40, 256, 3, 1, 2, 2),     # dilated 3x3
    (1, 128, 33, 47, 96, 3, 1, 1, 1),      # odd sizes: M tail and Cout tail (96)
    (2, 128, 32, 48, 128, 3, 2, 1, 1),     # stride 2
    (1, 512, 12, 16, 64, 1, 1, 0, 1),      # M = 192: split-K plan (atomics: compared with a tolerance)
    (2, 32, 40, 48, 64, 3, 1, 4, 4),       # Cin = 32: one K step per tap
]


@pytest.mark.parametrize("tile", [(0, 0), (1, 1), (1, 2), (2, 1), (2, 2), (2, 3)], ids=str)
@pytest.mark.parametrize("case", X3_SCHEDULE_CASES, ids=str)
def test_bf16x3_schedules_agree_bitwise(dev, case, tile):
    """The wave-specialised schedule (variant field bk = 2) computes every output element with the same MFMA sequence as the
    two-phase schedule (bk = 1): bit-identical outputs and double-precision batch statistics equal to rounding of the
    atomics' order, for every tile shape."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=41)
    wt = rnd(cout, cin, k, k, seed=42, scale=1.0 / np.sqrt(cin * k * k))
    b = rnd(cout, seed=43)
    tm, tn = tile
    if tn == 3:  # the eight-wave 128 x 256 tile of the first schedule against its own 128 x 128 tile
        y1, s1 = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(1, 2, 2, 2), bias=b, want_stats=True)
        y2, s2 = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(1, 2, 3, 2), bias=b, want_stats=True)
    else:
        y1, s1 = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(1, tm, tn, 2), bias=b, want_stats=True)
        y2, s2 = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(2, tm, tn, 2), bias=b, want_stats=True)
    npix = y1.shape[0] * y1.shape[2] * y1.shape[3]
    if npix > 2048:
        assert torch.equal(y1, y2)
    else:  # split-K: float atomics, order-dependent in the last bits
        close(y2, y1, atol=1e-5, what="split-K")
    t1, t2 = s1.view(ops.STAT_REPLICAS, 2, cout).sum(0).cpu(), s2.view(ops.STAT_REPLICAS, 2, cout).sum(0).cpu()  # doubles
    # (per-lane fp32 partial sums over <= 64 rows: the two kernels may contract v*v + s differently)
    assert float((t1 - t2).abs().max()) <= 1e-6 * float(t1.abs().max()), "batch statistics differ between the schedules"
    # and both are the convolution
    yr = F.conv2d(x, wt, b, stride=stride, padding=pad, dilation=dil)
    close(y2, yr, atol=(3e-6 * np.sqrt(cin * k * k) + 2e-6) * 4, what="y vs torch")


@pytest.mark.parametrize("case", X3_SCHEDULE_CASES, ids=str)
def test_bf16x3_schedules_agree_through_autograd(dev, case):
    """Same comparison for the data gradient (incl. the stride-2 parity classes), through ops.conv2d with the schedule
    forced by the tune-cache-free variant path."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=44)
    wt = rnd(cout, cin, k, k, seed=45, scale=1.0 / np.sqrt(cin * k * k))
    outs = []
    prev_math = ops.set_conv_math("bf16x3")
    prev_tune = ops.AUTOTUNE
    cands = ops._igemm_candidates
    try:
        ops.AUTOTUNE = False  # candidates[0] is used
        for bk in (1, 2):
            ops._igemm_candidates = lambda cout_, kind, cred=0, math_name=None, bk=bk: [ops._variant(bk, 0, 0, 2)]
            xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            y = ops.conv2d(xd, wd, None, stride, pad, dil)
            y.backward(rnd(*y.shape, seed=46).to(dev))
            outs.append((y.detach().clone(), xd.grad.detach().clone()))
    finally:
        ops._igemm_candidates = cands
        ops.AUTOTUNE = prev_tune
        ops.set_conv_math(prev_math)
    (y1, dx1), (y2, dx2) = outs
    npix_y = y1.shape[0] * y1.shape[2] * y1.shape[3]
    npix_x = n * h * w
    if npix_y > 2048:
        assert torch.equal(y1, y2)
    if npix_x > 2048 * (4 if stride == 2 else 1):
        assert torch.equal(dx1, dx2)
    else:
        close(dx2, dx1, atol=1e-5, what="dx (split-K)")


@pytest.mark.parametrize("case", X3_SCHEDULE_CASES + [(2, 64, 40, 44, 64, 3, 1, 1, 1), (2, 128, 24, 40, 128, 1, 1, 0, 1)], ids=str)
def test_conv2d_bf16_arithmetic(dev, case):
    """HN_MATH_BF16 (configs[4]): the result must be the convolution of the bf16-ROUNDED operands accumulated in fp32 --
    checked against exactly that computed in fp64 (so the only differences are fp32 accumulation order), and its distance
    from the unrounded fp32 convolution must be what 8-bit significands give (relative 2^-8 / sqrt(3) per operand)."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=51)
    wt = rnd(cout, cin, k, k, seed=52, scale=1.0 / np.sqrt(cin * k * k))
    gy_seed = 53
    xb, wb = x.bfloat16().double().requires_grad_(), wt.bfloat16().double().requires_grad_()
    yr = F.conv2d(xb, wb, None, stride=stride, padding=pad, dilation=dil)
    gy = rnd(*yr.shape, seed=gy_seed)
    # the gradients' operand dY is rounded too (it is the gathered operand of dgrad and one operand of wgrad)
    yr.backward(gy.bfloat16().double())
    prev = ops.set_conv_math("bf16")
    try:
        xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        y = ops.conv2d(xd, wd, None, stride, pad, dil)
        y.backward(gy.to(dev))
    finally:
        ops.set_conv_math(prev)
    kdim = cin * k * k
    tol = 3e-6 * np.sqrt(kdim) + 2e-6
    close(y, yr, atol=tol * 4, what="y vs bf16-rounded fp64")
    # dx: the kernel multiplies bf16(dY) with bf16(W): the same reference
    close(xd.grad, xb.grad, atol=(3e-6 * np.sqrt(cout * k * k) + 2e-6) * 4, what="dx vs bf16-rounded fp64")
    npix = yr.shape[0] * yr.shape[2] * yr.shape[3]
    if cout % 64 == 0 and cin % 64 == 0:
        close(wd.grad, wb.grad, atol=2e-5 * np.sqrt(npix) + 1e-5, rtol=1e-4, what="dw vs bf16-rounded fp64")
    else:  # the bf16 weight-gradient kernel needs whole 64-channel blocks of dY and of X: this one runs on the fp32 MFMA
        xr32, wr32 = x.double().requires_grad_(), wt.double().requires_grad_()
        F.conv2d(xr32, wr32, None, stride=stride, padding=pad, dilation=dil).backward(gy.double())
        close(wd.grad, wr32.grad, atol=2e-5 * np.sqrt(npix) + 1e-5, rtol=1e-4, what="dw (fp32 fallback) vs fp64")
    # distance from the fp32 convolution: rounding noise of two 8-bit operands, ~ 2^-9 * sqrt(2 K) * rms(x) * rms(w)
    y32 = F.conv2d(x, wt, None, stride=stride, padding=pad, dilation=dil)
    rel = float((y.detach().cpu() - y32).pow(2).mean().sqrt() / y32.pow(2).mean().sqrt())
    assert 5e-4 < rel < 8e-3, rel


AFFINE_CASES = [(2, 64, 40, 44, 64, 3, 1, 1, 1), (2, 256, 32, 40, 256, 3, 1, 2, 2), (2, 128, 32, 40, 512, 1, 1, 0, 1),
                (2, 128, 33, 47, 96, 3, 1, 4, 4), (2, 128, 64, 96, 128, 3, 2, 1, 1)]


@pytest.mark.parametrize("math", ["bf16x3", "bf16"])
@pytest.mark.parametrize("case", AFFINE_CASES, ids=str)
def test_conv2d_affine_loader(dev, case, math):
    """hn_conv2d_fwd_affine == hn_conv2d_fwd on the materialised relu(x * scale + shift): bit-identical (the transform is
    the same two fp32 operations, fmaf then max, and taps in the zero padding stay zero although relu(shift) != 0)."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=61)
    wt = rnd(cout, cin, k, k, seed=62, scale=1.0 / np.sqrt(cin * k * k))
    sc, sh = rnd(cin, seed=63) * 0.5 + 1.0, rnd(cin, seed=64) * 0.5 + 0.3  # mostly positive shifts: padding must stay 0
    # the kernel's transform is fmaf(x, scale, shift) then max(., 0): one rounding -- the same value from double arithmetic
    z = torch.relu((x.double() * sc.double().view(1, -1, 1, 1) + sh.double().view(1, -1, 1, 1)).float())
    var = ops._variant(2 if math == "bf16x3" else 0, 0, 0, ops._MATH_CODES[math])  # (bf16x3: the conv_igemm2 schedule)
    y_ref, s_ref = _abi_conv(dev, z, wt, stride, pad, dil, var, want_stats=True)
    y_aff, s_aff = _abi_conv(dev, x, wt, stride, pad, dil, var, want_stats=True, affine=(sc, sh))
    npix = y_ref.shape[0] * y_ref.shape[2] * y_ref.shape[3]
    assert npix > 2048
    # bit-identical up to the (very rare) double rounding of the host-side z; under bf16 such a one-ulp difference of z can
    # flip a bf16 rounding (2^-9 relative on one product)
    close(y_aff, y_ref, atol=1e-6 if math == "bf16x3" else 2e-3, what="affine vs materialised")
    yr = F.conv2d(z, wt, None, stride=stride, padding=pad, dilation=dil)
    tol = (3e-6 * np.sqrt(cin * k * k) + 2e-6) * 4 if math == "bf16x3" else 2e-2
    close(y_aff, yr, atol=tol, what="affine vs torch")


def test_conv2d_affine_is_refused_where_no_kernel_has_it(dev):
    import ctypes
    from heatnet_amd import _lib
    ops = _ops()
    x, wt = rnd(2, 64, 40, 44, seed=1), rnd(64, 64, 3, 3, seed=2)
    with pytest.raises(RuntimeError, match="hn_conv2d_fwd_affine"):
        _abi_conv(dev, x, wt, 1, 1, 1, ops._variant(0, 0, 0, 1), affine=(torch.ones(64), torch.zeros(64)))  # fp32 MFMA


def test_deterministic_mode_is_bitwise_reproducible(dev):
    """HEATNET_DETERMINISTIC: weight gradients through hn_conv2d_wgrad_det (per-split slabs + fixed-order sum) and no
    split-K in forward / data gradient: two runs agree bit for bit, also for a layer applied twice (accumulate), and the
    result equals the default (atomic) path to rounding."""
    ops = _ops()
    from heatnet_amd import optim
    import torch.nn as nn
    from heatnet_amd.layers import HnConv2d
    cases = [(2, 64, 40, 44, 64, 3, 1, 1, 1), (1, 512, 12, 16, 64, 1, 1, 0, 1), (2, 128, 32, 48, 128, 3, 2, 1, 1),
             (2, 3, 32, 48, 64, 7, 2, 3, 1)]
    for math in ("bf16x3", "f32"):
        prev_math = ops.set_conv_math(math)
        try:
            for n, cin, h, w, cout, k, stride, pad, dil in cases:
                x1, x2 = rnd(n, cin, h, w, seed=71).to(dev), rnd(n, cin, h, w, seed=72).to(dev)
                conv = HnConv2d(cin, cout, k, stride=stride, padding=pad, dilation=dil, bias=False).to(dev)
                arena = optim.ParamArena(list(conv.parameters()))
                runs = {}
                for mode in ("det", "det", "atomic"):
                    prev = ops.set_deterministic(mode == "det")
                    try:
                        arena.zero_grad()
                        a = x1.clone().contiguous(memory_format=torch.channels_last).requires_grad_()
                        b = x2.clone().contiguous(memory_format=torch.channels_last).requires_grad_()
                        (conv(a).square().mean() + conv(b).square().mean()).backward()  # the layer is used twice
                        res = (arena.grad.clone(), a.grad.clone(), b.grad.clone())
                    finally:
                        ops.set_deterministic(prev)
                    if mode in runs:
                        for t0, t1 in zip(runs[mode], res):
                            assert torch.equal(t0, t1), (math, (n, cin, h, w, cout, k), "deterministic runs differ")
                    runs[mode] = res
                for t0, t1 in zip(runs["det"], runs["atomic"]):
                    scale = float(t1.abs().max()) + 1e-30
                    assert float((t0 - t1).abs().max()) <= 2e-5 * scale
        finally:
            ops.set_conv_math(prev_math)


def test_weighted_cross_entropy_and_teacher_certainty(dev):
    """hn_teacher_cert / hn_ce_weighted_* against the reference's expressions (cm/train_trgb_segnet_conf.py:430-434,
    468-477): probs = softmax(teacher logits); label = argmax probs; cert = softmax(probs).max(1)[0];
    loss = mean(cert * CrossEntropyLoss(reduction='none')(logits, label))."""
    ops = _ops()
    t_logits = rnd(2, 13, 24, 40, seed=81, scale=2.0)
    logits = rnd(2, 13, 24, 40, seed=82)
    probs = F.softmax(t_logits, dim=1)
    label_ref = torch.argmax(probs, 1)
    cert_ref = F.softmax(probs, dim=1).max(1)[0]
    lr = logits.clone().requires_grad_()
    loss_ref = torch.mean(cert_ref * F.cross_entropy(lr, label_ref, reduction="none"))
    loss_ref.backward()
    label, cert = ops.teacher_outputs(t_logits.to(dev).contiguous(memory_format=torch.channels_last))
    assert float((label.cpu() != label_ref).float().mean()) == 0.0
    close(cert, cert_ref, atol=1e-6, what="cert")
    ld = logits.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    loss = ops.cross_entropy_weighted(ld, label, cert)
    loss.backward()
    assert abs(loss.item() - loss_ref.item()) < 1e-5
    close(ld.grad, lr.grad, atol=1e-7, rtol=1e-4, what="dlogits")


def test_dropout_mask_and_layout_conversion(dev):
    """hn_dropout_mask: factors in {0, 1/(1-p)}, keep rate ~ 1-p, reproducible from torch's CPU seed; hn_nchw_to_nhwc: a
    plain NCHW batch reaches the kernels through our own transposing kernel, values unchanged."""
    ops = _ops()
    torch.manual_seed(123)
    m1 = ops.drop_mask(64, 1024, 0.3, dev)
    torch.manual_seed(123)
    m2 = ops.drop_mask(64, 1024, 0.3, dev)
    assert torch.equal(m1, m2)
    vals = torch.unique(m1).cpu()
    assert vals.numel() == 2 and float(vals[0]) == 0.0 and abs(float(vals[1]) - 1.0 / 0.7) < 1e-6
    keep = float((m1 > 0).float().mean())
    assert abs(keep - 0.7) < 0.01, keep
    assert abs(float((m1 > 0).float().mean(1).std()) - np.sqrt(0.21 / 1024)) < 0.005  # rows are independent draws
    x = rnd(3, 5, 7, 11, seed=90).to(dev)  # NCHW-contiguous, odd sizes
    t, ld = ops.as_nhwc(x)
    assert ld == 5 and t.stride(1) == 1 and torch.equal(t.contiguous(), x)
    x = rnd(2, 64, 33, 47, seed=91).to(dev)
    t, ld = ops.as_nhwc(x)
    assert torch.equal(t.contiguous(), x)


@pytest.mark.parametrize("case", [(2, 64, 40, 44, 256, 1, 1, 0, 1), (2, 256, 32, 40, 256, 3, 1, 2, 2), (2, 128, 33, 47, 320, 3, 1, 1, 1),
                                  (3, 128, 64, 96, 512, 3, 2, 1, 1), (2, 32, 40, 48, 192, 3, 1, 4, 4)], ids=str)  # > 2048 output pixels: no split-K
def test_bf16x3_lds_dma_schedule_agrees_bitwise(dev, case):
    """Third schedule (variant bk = 3: 128 x 256 tile, weight planes by LDS-DMA into a double-buffered stage, raw barriers)
    against the first: bit-identical outputs incl. ragged M / Cout tails, bias, statistics."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=47)
    wt = rnd(cout, cin, k, k, seed=48, scale=1.0 / np.sqrt(cin * k * k))
    b = rnd(cout, seed=49)
    y1, s1 = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(1, 2, 2, 2), bias=b, want_stats=True)
    for _ in range(3):  # (repeat: a missing wait would show up as an occasional stale tile)
        y3, s3 = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(3, 0, 0, 2), bias=b, want_stats=True)
        assert torch.equal(y1, y3)
    t1, t3 = s1.view(ops.STAT_REPLICAS, 2, cout).sum(0).cpu(), s3.view(ops.STAT_REPLICAS, 2, cout).sum(0).cpu()
    assert float((t1 - t3).abs().max()) <= 1e-6 * float(t1.abs().max())


# ---- f16x2: operand preparation, scales, the *_scaled entry points through the C ABI --------------------------------------

def _slot_max(slot):
    """largest magnitude recorded in a scale slot (the maximum over its words, as a float)"""
    return float(slot.view(torch.int32).max().view(torch.float32))


def _expected_scale(amax):
    import math
    return 1.0 if (amax == 0.0 or not math.isfinite(amax)) else 2.0 ** (14 - math.floor(math.log2(amax)))


def test_f16x2_absmax_scale_slots(dev):
    """hn_absmax_scale: the slot receives the exact bit pattern of max|x| (dense and channel-slice inputs, negative extreme),
    ACCUMULATES over calls, and the BatchNorm passes publish the same value for their outputs."""
    import ctypes
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    x = rnd(2, 37, 41, 64, seed=70).to(dev)  # N H W C
    x[1, 5, 7, 9] = -123.456
    slot = torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev)
    _lib.check(lib.hn_absmax_scale(P(x), 64, 2 * 37 * 41, 64, P(slot), s), "absmax")
    assert _slot_max(slot) == float(x.abs().max()) == float(np.float32(123.456))
    sl = x[..., 8:40]  # channel slice: ld 64, C 32 (contains the extreme)
    slot2 = torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev)
    _lib.check(lib.hn_absmax_scale(P(sl), 64, 2 * 37 * 41, 32, P(slot2), s), "absmax")
    assert _slot_max(slot2) == float(sl.abs().max())
    small = (x * 1e-3).contiguous()
    _lib.check(lib.hn_absmax_scale(P(small), 64, 2 * 37 * 41, 64, P(slot2), s), "absmax")  # accumulates: the larger stays
    assert _slot_max(slot2) == float(sl.abs().max())
    # producers: z = relu(bn(y)) and the BatchNorm backward's dy carry their scale with them
    prev = ops.set_conv_math("f16x2")
    try:
        y = rnd(2, 64, 20, 24, seed=71).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
        g, b = (1.0 + 0.1 * rnd(64, seed=72)).to(dev).requires_grad_(), rnd(64, seed=73).to(dev).requires_grad_()
        z = ops.bn_act(y, None, g, b, torch.zeros(64, device=dev), torch.ones(64, device=dev), training=True, act=ops.ACT_RELU)
        assert _slot_max(z._hn_scale[1]) == float(z.detach().abs().max())
        before = ops.absmax_launches
        assert ops.tensor_scale(z) is z._hn_scale[1] and ops.absmax_launches == before  # remembered, no pass
        p = ops.maxpool3x3s2(z)
        assert p._hn_scale[1] is z._hn_scale[1]  # a bound: max pooling cannot exceed its input
        zd = z.detach().clone()
        zd._hn_scale = (zd._version, z._hn_scale[1])
        zd.add_(1.0)                             # in-place change: the remembered scale no longer applies
        assert ops.tensor_scale(zd) is not z._hn_scale[1] and ops.absmax_launches == before + 1
        assert _slot_max(zd._hn_scale[1]) == float(zd.abs().max())
    finally:
        ops.set_conv_math(prev)


@pytest.mark.parametrize("scale", [1.0, 3e-7, 4e5])
def test_f16x2_weight_planes(dev, scale):
    """hn_split_f16x2 / hn_weight_flip_transpose_h2 / the batched preparation: hi + lo reproduces w * s to 2^-22 of the
    tensor's maximum, s is the power of two that puts max|w| into [2^14, 2^15), all three routes give identical planes."""
    import ctypes
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    cout, kh, kw, cin = 96, 3, 3, 64
    w = (rnd(cout, kh, kw, cin, seed=74) * scale).to(dev)
    nw, plane = w.numel(), ops.x3_plane(w.numel())
    h2 = torch.empty(3 * plane, device=dev, dtype=torch.int16)
    _lib.check(lib.hn_split_f16x2(P(w), P(h2), nw, s), "split")
    area = h2[2 * plane:].view(torch.float32)
    sc, inv = float(area[0]), float(area[1])
    assert sc == _expected_scale(float(w.abs().max())) and sc * inv == 1.0
    hi, lo = h2[:nw].view(torch.float16).double(), h2[plane:plane + nw].view(torch.float16).double()
    ws = w.double().flatten() * sc
    assert 2.0 ** 14 <= float(ws.abs().max()) < 2.0 ** 15
    assert float((hi + lo - ws).abs().max()) <= 2.0 ** -22 * float(ws.abs().max())
    # the flip / transpose route: same forward planes, and the transposed planes hold the same values re-ordered
    w2, wt, wt2 = torch.empty_like(h2), torch.empty(nw, device=dev), torch.empty_like(h2)
    _lib.check(lib.hn_weight_flip_transpose_h2(P(w), P(w2), P(wt), P(wt2), cout, kh, kw, cin, 1, 1, 1, s), "flip")
    assert torch.equal(w2[:2 * plane], h2[:2 * plane]) and torch.equal(w2[2 * plane:2 * plane + 4], h2[2 * plane:2 * plane + 4])
    ref_t = w.flip(1, 2).permute(3, 1, 2, 0).contiguous().flatten()  # [Cin][KH][KW][Cout], taps flipped
    assert torch.equal(wt, ref_t)
    t_hi, t_lo = wt2[:nw].view(torch.float16).double(), wt2[plane:plane + nw].view(torch.float16).double()
    assert float((t_hi + t_lo - ref_t.double() * sc).abs().max()) <= 2.0 ** -22 * float(ws.abs().max())
    assert float(wt2[2 * plane:].view(torch.float32)[0]) == sc
    # batched: one table with an f16x2 and a bf16x3 layer
    w3 = torch.empty_like(h2)
    wt3, wt4 = torch.empty(nw, device=dev), torch.empty_like(h2)
    wb = rnd(64, 1, 1, 32, seed=75).to(dev)
    xb, xb_ref = torch.empty(3 * ops.x3_plane(wb.numel()), device=dev, dtype=torch.int16), None
    items = (_lib.HnWeightPrepItem * 2)()
    items[0].w, items[0].w_x3, items[0].wt, items[0].wt_x3 = w.data_ptr(), w3.data_ptr(), wt3.data_ptr(), wt4.data_ptr()
    items[0].Cout, items[0].KH, items[0].KW, items[0].Cin, items[0].stride, items[0].pad, items[0].dil, items[0].format = cout, kh, kw, cin, 1, 1, 1, 1
    items[1].w, items[1].w_x3, items[1].wt, items[1].wt_x3 = wb.data_ptr(), xb.data_ptr(), None, None
    items[1].Cout, items[1].KH, items[1].KW, items[1].Cin, items[1].stride, items[1].pad, items[1].dil, items[1].format = 64, 1, 1, 32, 1, 0, 1, 0
    host = torch.empty(lib.hn_weight_prep_table_bytes(2), dtype=torch.uint8)
    blocks = ctypes.c_int(0)
    _lib.check(lib.hn_weight_prep_table_fill(items, 2, ctypes.c_void_p(host.data_ptr()), ctypes.byref(blocks)), "fill")
    table = host.to(dev)
    _lib.check(lib.hn_weight_prepare_batch(P(table), 2, blocks.value, 1, s), "batch")
    assert torch.equal(w3[:2 * plane], h2[:2 * plane]) and torch.equal(wt4[:2 * plane], wt2[:2 * plane]) and torch.equal(wt3, wt)
    assert float(w3[2 * plane:].view(torch.float32)[0]) == sc and float(wt4[2 * plane:].view(torch.float32)[0]) == sc
    xb_ref = torch.empty_like(xb)
    _lib.check(lib.hn_split_bf16x3(P(wb), P(xb_ref), wb.numel(), s), "split3")
    assert torch.equal(xb, xb_ref)


@pytest.mark.parametrize("mag", [(1.0, 1.0), (1e-12, 1e9), (3e7, 1e-20)], ids=str)
def test_f16x2_scaled_entry_points(dev, mag):
    """hn_conv2d_{fwd,dgrad,wgrad}_scaled through the C ABI on operands of very different magnitudes (the power-of-two
    scales keep fp16 in range): against an fp64 convolution, as accurate as bf16x3 (relative to the result's own rms);
    bias / ReLU / statistics epilogue, skip-gradient addend, deterministic weight gradient.  Refusals: no scale, planes of
    the wrong call, uncovered geometry."""
    import ctypes
    from heatnet_amd import _lib, ops
    lib = _lib.load()
    P = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None else None
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    xm, wm = mag
    n, cin, h, w_, cout, k, pad = 2, 128, 32, 40, 192, 3, 1  # (> 2048 pixels: no split-K, bitwise repeatable)
    x = (F.relu(rnd(n, cin, h, w_, seed=76)) * xm)
    wt = rnd(cout, cin, k, k, seed=77, scale=wm / np.sqrt(cin * k * k))
    b = rnd(cout, seed=78) * (xm * wm)
    gy = rnd(n, cout, h, w_, seed=79) * 1e-4
    skip = rnd(n, cin, h, w_, seed=80) * (1e-4 * wm)
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    yr = F.conv2d(xr, wr, b.double(), padding=pad)
    yr.backward(gy.double())
    xd, wd = x.to(dev).permute(0, 2, 3, 1).contiguous(), wt.to(dev).permute(0, 2, 3, 1).contiguous()
    gyd, skd, bd = gy.to(dev).permute(0, 2, 3, 1).contiguous(), skip.to(dev).permute(0, 2, 3, 1).contiguous(), b.to(dev)
    nw, plane = wd.numel(), ops.x3_plane(wd.numel())
    wh, wth, wtf = (torch.empty(3 * plane, device=dev, dtype=torch.int16), torch.empty(3 * plane, device=dev, dtype=torch.int16),
                    torch.empty(nw, device=dev))
    _lib.check(lib.hn_weight_flip_transpose_h2(P(wd), P(wh), P(wtf), P(wth), cout, k, k, cin, 1, pad, 1, s), "flip")
    xs, gs = torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev), torch.zeros(ops.SCALE_SLOT_FLOATS, device=dev)
    _lib.check(lib.hn_absmax_scale(P(xd), cin, n * h * w_, cin, P(xs), s), "absmax")
    _lib.check(lib.hn_absmax_scale(P(gyd), cout, n * h * w_, cout, P(gs), s), "absmax")
    d = _lib.HnConvDesc(n, h, w_, cin, cin, h, w_, cout, cout, k, k, 1, pad, 1)
    y = torch.empty(n, h, w_, cout, device=dev)
    stats = torch.zeros(ops.STAT_REPLICAS * 2 * cout, device=dev, dtype=torch.float64)
    _lib.check(lib.hn_conv2d_fwd_scaled(P(xd), P(xs), P(wd), P(wh), P(bd), P(y), P(stats), ctypes.byref(d), 0, 0.0, 0, s), "fwd")
    dx = torch.empty_like(xd)
    _lib.check(lib.hn_conv2d_dgrad_scaled(P(gyd), P(gs), P(wtf), P(wth), P(dx), P(skd), cin, ctypes.byref(d), 0, s), "dgrad")
    dw, dw_det = torch.empty_like(wd), torch.empty_like(wd)
    _lib.check(lib.hn_conv2d_wgrad_scaled(P(xd), P(xs), P(gyd), P(gs), P(dw), ctypes.byref(d), 0, 0, 0, s), "wgrad")
    wsp = torch.empty(4 * nw, device=dev)
    _lib.check(lib.hn_conv2d_wgrad_det_scaled(P(xd), P(xs), P(gyd), P(gs), P(dw_det), ctypes.byref(d), 4, 0, 0, P(wsp), s), "wgrad_det")

    def rel(a, ref):
        ref = ref.detach()
        return float((a.double().cpu() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())
    assert rel(y.permute(0, 3, 1, 2), yr) < 4e-7
    assert rel(dx.permute(0, 3, 1, 2), xr.grad + skip.double()) < 6e-7
    assert rel(dw.permute(0, 3, 1, 2), wr.grad) < 4e-7 and rel(dw_det.permute(0, 3, 1, 2), wr.grad) < 4e-7
    st = stats.view(ops.STAT_REPLICAS, 2, cout).sum(0).cpu()
    ref_sum = yr.detach().sum((0, 2, 3))
    assert float((st[0] - ref_sum).abs().max()) <= 1e-5 * float(yr.detach().abs().sum((0, 2, 3)).max())
    # epilogue activation
    y2 = torch.empty_like(y)
    _lib.check(lib.hn_conv2d_fwd_scaled(P(xd), P(xs), P(wd), P(wh), P(bd), P(y2), None, ctypes.byref(d), 1, 0.0, 0, s), "fwd relu")
    assert torch.equal(y2, torch.relu(y))
    # refusals
    assert lib.hn_conv2d_fwd_scaled(P(xd), None, P(wd), P(wh), None, P(y), None, ctypes.byref(d), 0, 0.0, 0, s) == -1  # HN_EINVAL
    assert lib.hn_conv2d_fwd(P(xd), P(wd), P(wh), None, P(y), None, ctypes.byref(d), 0, 0.0, ops._variant(0, 0, 0, 4), s) == -1
    d3 = _lib.HnConvDesc(n, h, w_, 3, cin, h, w_, cout, cout, k, k, 1, pad, 1)  # Cin = 3: not an f16x2 geometry
    assert lib.hn_conv2d_fwd_scaled(P(xd), P(xs), P(wd), P(wh), None, P(y), None, ctypes.byref(d3), 0, 0.0, 0, s) == -4  # HN_ESHAPE
    d13 = _lib.HnConvDesc(n, h, w_, cin, cin, h, w_, 13, cout, k, k, 1, pad, 1)  # Cout = 13: no f16x2 weight gradient
    assert lib.hn_conv2d_wgrad_scaled(P(xd), P(xs), P(gyd), P(gs), P(dw), ctypes.byref(d13), 0, 0, 0, s) == -4


def test_f16x2_all_zero_and_nonfinite_operands(dev):
    """An all-zero tensor has scale 1 and an exactly zero result; an infinite element makes the outputs it reaches
    non-finite, as under fp32 (the tensor's scale then falls back to 1: no statement about the other outputs)."""
    ops = _ops()
    prev = ops.set_conv_math("f16x2")
    try:
        w = rnd(64, 64, 3, 3, seed=81).to(dev).contiguous(memory_format=torch.channels_last)
        x0 = torch.zeros(1, 64, 16, 24, device=dev).contiguous(memory_format=torch.channels_last)
        assert float(ops.conv2d(x0, w, None, 1, 1, 1).abs().max()) == 0.0
        x1 = rnd(1, 64, 16, 24, seed=82).to(dev).contiguous(memory_format=torch.channels_last)
        x1[0, 3, 4, 5] = float("inf")
        y = ops.conv2d(x1, w, None, 1, 1, 1)
        assert not torch.isfinite(y[0, :, 4, 5]).all()
    finally:
        ops.set_conv_math(prev)


@pytest.mark.parametrize("case", X3_SCHEDULE_CASES + [(2, 128, 40, 44, 256, 3, 1, 1, 1)], ids=str)
def test_f16x2_schedules_agree_bitwise(dev, case):
    """The other schedules of the f16x2 arithmetic -- wave-specialised (variant bk = 2: producer / consumer waves, K step 64
    where Cin allows) and weights-by-LDS-DMA (bk = 3: 128 x 128 with two workgroups per CU, or tn = 3: 128 x 256) -- against
    the two-phase one: bit-identical outputs through ops.conv2d, forward and data gradient (split-K cases with a tolerance:
    float atomics; geometries a schedule does not cover run the two-phase kernel)."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=84)
    wt = rnd(cout, cin, k, k, seed=85, scale=1.0 / np.sqrt(cin * k * k))
    outs = []
    prev_math = ops.set_conv_math("f16x2")
    prev_tune = ops.AUTOTUNE
    cands = ops._igemm_candidates
    try:
        ops.AUTOTUNE = False  # candidates[0] is used
        for bk, tn in ((0, 0), (2, 0), (3, 0), (3, 3)):  # two-phase | wave-specialised | LDS-DMA 128x128 | LDS-DMA 128x256
            ops._igemm_candidates = lambda cout_, kind, cred=0, math_name=None, bk=bk, tn=tn: [ops._variant(bk, 0, tn, ops._MATH_CODES[math_name])]
            xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
            y = ops.conv2d(xd, wd, None, stride, pad, dil)
            y.backward(rnd(*y.shape, seed=86).to(dev))
            outs.append((y.detach().clone(), xd.grad.detach().clone()))
    finally:
        ops._igemm_candidates = cands
        ops.AUTOTUNE = prev_tune
        ops.set_conv_math(prev_math)
    y1, dx1 = outs[0]
    for y2, dx2 in outs[1:]:
        if y1.shape[0] * y1.shape[2] * y1.shape[3] > 2048:
            assert torch.equal(y1, y2)
        else:
            close(y2, y1, atol=1e-5, what="y (split-K)")
        if n * h * w > 2048 * (4 if stride == 2 else 1):
            assert torch.equal(dx1, dx2)
        else:
            close(dx2, dx1, atol=1e-5, what="dx (split-K)")


@pytest.mark.parametrize("case", [(2, 256, 32, 40, 256, 3, 1, 2, 2), (2, 64, 40, 44, 256, 1, 1, 0, 1), (2, 128, 40, 44, 128, 3, 1, 1, 1)], ids=str)  # > 2048 pixels: no split-K
def test_bf16_schedules_agree_bitwise(dev, case):
    """The reduced-precision bf16 arithmetic on the LDS-DMA schedule (variant bk = 3, both tiles) against the wave-specialised
    kernel it normally runs on: bit-identical outputs (same rounding of the operands, same accumulation order)."""
    ops = _ops()
    n, cin, h, w, cout, k, stride, pad, dil = case
    x = rnd(n, cin, h, w, seed=87)
    wt = rnd(cout, cin, k, k, seed=88, scale=1.0 / np.sqrt(cin * k * k))
    y0, _ = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(0, 0, 0, 3))
    for tn in (0, 3):
        y3, _ = _abi_conv(dev, x, wt, stride, pad, dil, ops._variant(3, 0, tn, 3))
        assert torch.equal(y0, y3)


@pytest.mark.parametrize("c", [64, 6])
def test_bn_relu_byte_mask_equals_reading_z(dev, c):
    """Residual + ReLU layers: the backward passes take the ReLU mask from one byte per four channels written by the forward
    pass (a 16th of z's bytes); gradients must be bitwise those of the path that reads z (C % 4 != 0 always reads z)."""
    ops = _ops()
    n, h, w = 2, 12, 16
    y, res = rnd(n, c, h, w, seed=95, scale=2.0), rnd(n, c, h, w, seed=96)
    gamma, beta = rnd(c, seed=97).abs() + 0.5, rnd(c, seed=98) * 0.1
    gz = rnd(n, c, h, w, seed=99)
    outs = []
    prev = ops.RELU_BITS
    try:
        for bits in (True, False):
            ops.RELU_BITS = bits
            cl = torch.channels_last
            yd = y.to(dev).contiguous(memory_format=cl).requires_grad_()
            rd = res.to(dev).contiguous(memory_format=cl).requires_grad_()
            gd, bd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
            z = ops.bn_act(yd, None, gd, bd, torch.zeros(c, device=dev), torch.ones(c, device=dev), residual=rd, training=True,
                           act=ops.ACT_RELU)
            saved = [t for t in z.grad_fn.saved_tensors if t is not None and t.dtype == torch.uint8]
            assert bool(saved) == (bits and c % 4 == 0)
            z.backward(gz.to(dev))
            outs.append((z.detach().clone(), yd.grad.clone(), rd.grad.clone(), gd.grad.clone(), bd.grad.clone()))
    finally:
        ops.RELU_BITS = prev
    for a, b in zip(*outs):
        assert torch.equal(a, b)


def test_bn_relu_byte_mask_takes_an_unaligned_gradient_view(dev):
    """The byte ReLU masks replace the saved z, and the kernels that read them work on whole 16-byte pieces: a gradient that
    reaches the backward as a channel slice of a wider buffer (pixel stride no multiple of four, base not 16-byte aligned)
    used to fail with HN_EALIGN -- it is re-laid out instead.  Same gradients as for a dense copy of it."""
    ops = _ops()
    n, c, h, w = 2, 64, 12, 16
    y, res = rnd(n, c, h, w, seed=195, scale=2.0), rnd(n, c, h, w, seed=196)
    gamma, beta = rnd(c, seed=197).abs() + 0.5, rnd(c, seed=198) * 0.1
    wide = rnd(n, c + 3, h, w, seed=199).to(dev).contiguous(memory_format=torch.channels_last)
    outs = []
    for view in (True, False):
        cl = torch.channels_last
        yd = y.to(dev).contiguous(memory_format=cl).requires_grad_()
        rd = res.to(dev).contiguous(memory_format=cl).requires_grad_()
        gd, bd = gamma.to(dev).requires_grad_(), beta.to(dev).requires_grad_()
        z = ops.bn_act(yd, None, gd, bd, torch.zeros(c, device=dev), torch.ones(c, device=dev), residual=rd, training=True,
                       act=ops.ACT_RELU)
        assert any(t is not None and t.dtype == torch.uint8 for t in z.grad_fn.saved_tensors)
        gz = wide[:, 1:1 + c]  # pixel stride c + 3, base 4 bytes past a 16-byte boundary
        assert gz.data_ptr() % 16 != 0 and ops.nhwc_ld(gz) == c + 3
        z.backward(gz if view else gz.contiguous(memory_format=cl))
        outs.append((yd.grad.clone(), rd.grad.clone(), gd.grad.clone(), bd.grad.clone()))
    for a, b in zip(*outs):
        assert torch.equal(a, b)


@pytest.mark.parametrize("act", ["none", "leaky"])
def test_overlap_knob_f16x2_weight_gradient(dev, act):
    """HEATNET_OVERLAP=1 (wgrad on a side stream beside dgrad) under f16x2: the operand scales the side stream reads -- found by
    hn_absmax_scale passes on the MAIN stream for gradients that do not come out of a BatchNorm backward -- are queued before
    the fork event (they were not; the weight gradient could read a zero / partial scale).  Same gradients as
    without the knob, over repeated runs with fresh scale slots."""
    ops = _ops()
    n, cin, h, w, cout, k = 2, 128, 40, 44, 192, 3
    x = rnd(n, cin, h, w, seed=91)
    wt = rnd(cout, cin, k, k, seed=92, scale=1.0 / np.sqrt(cin * k * k))
    gy = rnd(n, cout, h, w, seed=93) * 1e-5  # small gradients: a missing scale would flush them to fp16 subnormals
    prev_math = ops.set_conv_math("f16x2")
    prev_overlap = ops.OVERLAP
    results = {}
    try:
        for overlap in (False, True):
            ops.OVERLAP = overlap
            outs = []
            for rep in range(4 if overlap else 1):
                xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
                wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
                if act == "leaky":
                    y = ops.conv2d(xd, wd, None, 1, 1, 1, act=ops.ACT_LEAKY, slope=0.2)
                else:
                    y = ops.conv2d(xd, wd, None, 1, 1, 1)
                y.backward(gy.to(dev))
                torch.cuda.synchronize()
                outs.append((xd.grad.clone(), wd.grad.clone()))
            results[overlap] = outs
    finally:
        ops.OVERLAP = prev_overlap
        ops.set_conv_math(prev_math)
    dx0, dw0 = results[False][0]
    for dx1, dw1 in results[True]:
        assert torch.equal(dx0, dx1)
        close(dw1, dw0, atol=2e-5 * float(dw0.abs().max()), what="dw (float atomics)")
