"""The ResNeXt-50 + ASPP "custom" architecture (cm/models/trgb_segnet.py:242-402, the default `arch` of conv_segnet) on the HIP
path: its own operators (grouped 3x3 convolution, transposed convolution) against fp64 torch on the CPU, the network against
the fixtures made by the reference's module (tests/golden/make_golden.py --part resnext), the wrapper."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _seeded import seeded_inputs, seeded_state, tensor_digest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel(a, ref):
    return float((a.detach().double().cpu() - ref).abs().max() / ref.abs().max())


@pytest.mark.parametrize("case", [(128, 64, 1, 40, 44), (256, 64, 1, 24, 40), (512, 64, 2, 17, 23), (1024, 64, 4, 16, 24),
                                  (64, 4, 1, 9, 11)], ids=str)
def test_grouped_conv3x3(dev, case):
    from heatnet_amd import ops
    c, groups, dil, h, w_ = case
    n = 2
    x = rnd(n, c, h, w_, seed=1)
    wt = rnd(c, c // groups, 3, 3, seed=2, scale=1.0 / np.sqrt(9 * c / groups))
    gy = rnd(n, c, h, w_, seed=3)
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    yr = F.conv2d(xr, wr, None, padding=dil, dilation=dil, groups=groups)
    yr.backward(gy.double())
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = ops.grouped_conv3x3(xd, wd, groups, dil)
    y.backward(gy.to(dev))
    assert rel(y, yr.detach()) < 2e-6
    assert rel(xd.grad, xr.grad) < 2e-6
    assert rel(wd.grad, wr.grad) < 1e-5


@pytest.mark.parametrize("case", [(13, 13, 4, 2, 1, 16, 24), (13, 13, 8, 4, 2, 32, 48), (5, 7, 4, 2, 1, 9, 11), (1, 1, 4, 2, 1, 16, 24), (1, 1, 8, 4, 2, 12, 20)], ids=str)
def test_conv_transpose2d(dev, case):
    from heatnet_amd import ops
    cin, cout, k, stride, pad, h, w_ = case
    n = 2
    x = rnd(n, cin, h, w_, seed=4)
    wt = rnd(cin, cout, k, k, seed=5, scale=0.2)
    xr, wr = x.double().requires_grad_(), wt.double().requires_grad_()
    yr = F.conv_transpose2d(xr, wr, None, stride=stride, padding=pad)
    gy = rnd(*yr.shape, seed=6)
    yr.backward(gy.double())
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    wd = wt.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = ops.conv_transpose2d(xd, wd, stride, pad)
    assert tuple(y.shape) == tuple(yr.shape)
    y.backward(gy.to(dev))
    assert rel(y, yr.detach()) < 2e-6
    assert rel(xd.grad, xr.grad) < 2e-6
    assert rel(wd.grad, wr.grad) < 1e-5


def test_sigmoid_and_certainty_loss_match_torch(dev):
    """ops.sigmoid and ops.cert_loss (hn_sigmoid_*, hn_cert_loss_*) against the trainer's lines in stock fp64 torch on the CPU:
    10 * mean((cert_a - (1 - p_label)) ** 2) with the [N, 1, H, W] - [N, H, W] broadcast to [N, N, H, W] and the gradient that
    reaches the logits through the (not detached) target."""
    from heatnet_amd import ops
    from oracle import resnext_ref as R
    n, c, h, w_ = 3, 13, 20, 28
    logits, pre = rnd(n, c, h, w_, seed=41, scale=2.0), rnd(n, 1, h, w_, seed=42, scale=1.5)
    lab = torch.randint(0, c, (n, h, w_), generator=torch.Generator().manual_seed(43))
    lr, pr = logits.double().requires_grad_(), pre.double().requires_grad_()
    ref = R.cert_loss(lr, lab, torch.sigmoid(pr)) * 0.7
    ref.backward()
    ld = logits.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    pd = pre.to(dev).requires_grad_()
    out = ops.cert_loss(ld, lab.to(dev), ops.sigmoid(pd)) * 0.7
    out.backward()
    assert abs(float(out.detach()) - float(ref.detach())) < 1e-6 * max(1.0, abs(float(ref.detach())))
    assert rel(ld.grad, lr.grad) < 2e-5
    assert rel(pd.grad, pr.grad) < 2e-5


@pytest.mark.parametrize("tag,late", [("late", True), ("early", False), ("cert", False)])
def test_resnext_matches_reference_fixture(dev, golden_dir, tag, late):
    """Train-mode forward, cross-entropy and backward of the custom architecture against the reference's own module: logits
    within 1e-3 (the tolerance BASELINE.json states for the segmentation logits), feature samples, loss, the gradient norm of
    every parameter, gradient slices, BatchNorm running statistics."""
    from heatnet_amd import train_step
    from heatnet_amd.trgb_segnet import ResNeXt
    g = np.load(os.path.join(golden_dir, "resnext_%s_2x128x192.npz" % tag))
    net = ResNeXt(structure=[3, 4, 6, 3], input_channels=4, late_fusion=late, cert_branch=(tag == "cert"))
    assert list(net.state_dict().keys()) == list(g["keys"])
    net.load_state_dict(seeded_state(net, seed=21))
    net = net.to(dev).train()
    rgb, ir, lab = seeded_inputs(2, 128, 192, seed=23)
    seg, feats, cert = net(rgb.to(dev), ir.to(dev))
    assert (cert is None) == (tag != "cert") and len(feats) == 6
    loss = train_step.segmentation_loss(seg, lab.to(dev))
    if tag == "cert":  # the certainty head (:344-346, 383-387) and the trainer's certainty loss (cm/train_trgb_segnet_conf.py:482-491)
        from heatnet_amd import ops
        closs = ops.cert_loss(seg, lab.to(dev), cert)
        assert float((cert.detach().cpu()[..., ::2, ::2] - torch.from_numpy(g["cert_sub"])).abs().max()) < 1e-3
        assert abs(float(closs.detach()) - float(g["cert_loss"])) < 1e-4 * max(1.0, float(g["cert_loss"]))
        loss = loss + closs
    loss.backward()
    torch.cuda.synchronize()
    err = float((seg.detach().cpu()[..., ::2, ::2] - torch.from_numpy(g["logits_sub"])).abs().max())
    assert err < 1e-3, err
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-4
    for i, t in enumerate(feats[1:]):
        assert tuple(t.shape) == tuple(g["feat%d_shape" % i])
        _, s = tensor_digest(t.contiguous(), 31)
        np.testing.assert_allclose(s, g["feat%d_sample" % i], rtol=0, atol=1e-3)
    params = dict(net.named_parameters())
    floor = 1e-4 * float(g["grad_norms"].max())
    worst = (0.0, None)
    for k, ref in zip(g["grad_keys"], g["grad_norms"]):
        assert params[k].grad is not None, k
        e = abs(float(params[k].grad.double().norm()) - ref) / max(ref, floor)
        worst = max(worst, (e, k))
    assert worst[0] < 2e-2, worst
    for k in g.files:
        if k.startswith("grad|"):
            name = k[5:]
            gr = params[name].grad.detach().cpu().reshape(-1)
            gr = gr[:: max(1, gr.numel() // 4096)].double().numpy()
            ref = g[k].astype(np.float64)
            if np.linalg.norm(ref) == 0.0:  # (e.g. the corner taps of a 3x3 with dilation 24 on a 16 x 24 map: never in range)
                assert np.linalg.norm(gr) == 0.0, name
                continue
            cos = float((gr * ref).sum() / (np.linalg.norm(gr) * np.linalg.norm(ref) + 1e-300))
            assert cos > 0.999, (name, cos)
    sd = net.state_dict()
    for k in ("mod1.bn1_2.bn.running_mean", "mod5_1.block3.convs.bn3.bn.running_var", "aspp.conv5.bn1.bn.running_var",
              "fuse_seg.bn1.bn.running_mean"):
        np.testing.assert_allclose(sd[k].cpu().numpy(), g["after_" + k], rtol=1e-3, atol=1e-5)


def test_conv_segnet_custom_arch_runs_both_phases(dev):
    """conv_segnet(arch='custom') (cm/models/conf_segnet.py:38-43): the critics' input widths, the output dictionary, one
    train_critic and one train_seg iteration with finite losses and gradients on exactly the trainable half."""
    from heatnet_amd import conf_segnet, train_step
    model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="custom", late_fusion=True)
    assert [c.conv1.in_channels for c in model.critics] == [13, 768, 1024, 512, 512, 128]
    model = model.to(dev).train()
    rgb_d, ir_d, lab = [t.to(dev) for t in seeded_inputs(2, 256, 256, seed=31)]
    rgb_n, ir_n, _ = [t.to(dev) for t in seeded_inputs(2, 256, 256, seed=32)]
    for state in ("train_critic", "train_seg"):
        model.setPhase(state)
        opt = torch.optim.RMSprop([p for p in model.parameters()], lr=1e-5)
        losses = train_step.train_step(model, opt, [rgb_d, ir_d], [rgb_n, ir_n], lab, state)
        assert all(torch.isfinite(v) for v in losses.values()), losses
        seg_has = any(p.grad is not None and float(p.grad.abs().max()) > 0 for p in model.trgb_segnet.parameters())
        assert seg_has == (state == "train_seg")
    with torch.no_grad():
        out = model([rgb_d, ir_d], [rgb_n, ir_n])
    assert set(out) == {"critics_a", "critics_b", "pred_label_a", "pred_label_b", "cert_a", "cert_b", "inter_f_b"}
    assert tuple(out["pred_label_a"].shape) == (2, 13, 256, 256) and out["cert_a"] is None


def test_conv_segnet_cert_branch_iteration(dev):
    """conv_segnet(arch='custom', cert_branch=True) (early fusion): the output dictionary carries the certainty maps and a
    train_seg iteration adds the trainer's certainty loss (cm/train_trgb_segnet_conf.py:482-491) -- every parameter of the
    certainty head receives a gradient; late fusion fails in the head's first convolution exactly as the reference does (its
    fuseModule is sized for the 256-channel early-fusion skip tensor)."""
    from heatnet_amd import conf_segnet, train_step
    model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="custom", late_fusion=False,
                                    cert_branch=True).to(dev).train()
    rgb_d, ir_d, lab = [t.to(dev) for t in seeded_inputs(2, 256, 256, seed=33)]
    rgb_n, ir_n, _ = [t.to(dev) for t in seeded_inputs(2, 256, 256, seed=34)]
    model.setPhase("train_seg")
    opt = torch.optim.RMSprop([p for p in model.parameters()], lr=1e-5)
    losses = train_step.train_step(model, opt, [rgb_d, ir_d], [rgb_n, ir_n], lab, "train_seg")
    assert "cert" in losses and all(torch.isfinite(v) for v in losses.values()), losses
    head = [p for k, p in model.trgb_segnet.named_parameters() if "cert" in k or "final_conv2" in k]
    assert len(head) >= 8 and all(p.grad is not None and float(p.grad.abs().max()) > 0 for p in head)
    with torch.no_grad():
        out = model([rgb_d, ir_d], [rgb_n, ir_n])
    assert tuple(out["cert_a"].shape) == (2, 1, 256, 256) and float(out["cert_a"].min()) >= 0 and float(out["cert_a"].max()) <= 1
    late = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="custom", late_fusion=True,
                                   cert_branch=True).to(dev).train()
    with pytest.raises(RuntimeError):
        late([rgb_d, ir_d], [rgb_n, ir_n])
