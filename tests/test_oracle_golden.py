"""CPU: the oracle (oracle/heatnet_ref.py) against fixtures produced by the reference itself.

The fixtures were written by tests/golden/make_golden.py, which runs the
reference's model files on CPU.  Tolerances are fp32 noise of the same torch
build (4e-5 thread-count noise on logits).
"""
import os

import numpy as np
import pytest
import torch

from _seeded import seeded_inputs, seeded_state, set_dropout, tensor_digest
from oracle import heatnet_ref as R


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + ".npz"), allow_pickle=False)


def _check_feats(g, prefix, feats, stride, tol):
    for i, t in enumerate(feats):
        assert tuple(g["%s%d_shape" % (prefix, i)]) == tuple(t.shape)
        d, s = tensor_digest(t, stride)
        np.testing.assert_allclose(s, g["%s%d_sample" % (prefix, i)], rtol=0, atol=tol)
        np.testing.assert_allclose(d[1], g["%s%d_digest" % (prefix, i)][1], rtol=1e-5)


@pytest.mark.parametrize("tag,late", [("late", True), ("early", False)])
def test_segnet_small_matches_reference(golden_dir, tag, late):
    g = _load(golden_dir, "segnet_%s_2x64x96" % tag)
    net = R.RefPSPNet(backend="resnet50", late_fusion=late, in_channels=4)
    net.load_state_dict(seeded_state(net, seed=1))
    set_dropout(net, 0.0)
    rgb, ir, _ = seeded_inputs(2, 64, 96, seed=3)
    net.train()
    logits, feats, none = net(rgb, ir)
    assert none is None and len(feats) == 6 and feats[0] is logits
    np.testing.assert_allclose(logits.detach().numpy(), g["train_logits"], rtol=0, atol=2e-4)
    _check_feats(g, "train_feat", feats[1:], 97, 2e-4)
    sd = net.state_dict()
    for k in ("feats.bn1.running_mean", "feats.bn1.running_var", "feats.layer4.2.bn3.running_mean",
              "feats.layer4.2.bn3.running_var", "up_3.conv.1.running_var"):
        np.testing.assert_allclose(sd[k].numpy(), g["after_" + k], rtol=1e-5, atol=1e-6)
    assert int(sd["feats.bn1.num_batches_tracked"]) == int(g["after_nbt"])
    net.load_state_dict(seeded_state(net, seed=1))
    net.eval()
    with torch.no_grad():
        logits, feats, _ = net(rgb, ir)
    np.testing.assert_allclose(logits.numpy(), g["eval_logits"], rtol=0, atol=2e-4)
    _check_feats(g, "eval_feat", feats[1:], 97, 2e-4)


def test_state_dict_keys_match_reference_layout():
    net = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    sd = net.state_dict()
    assert len(sd) == 488
    assert sd["feats.conv1_2.weight"].shape == (64, 1, 7, 7)
    assert sd["psp.bottleneck.weight"].shape == (1024, 10240, 1, 1)
    assert sd["psp.stages.3.1.weight"].shape == (2048, 2048, 1, 1)
    assert sd["up_1.conv.2.weight"].shape == (1,)
    assert sd["final.0.weight"].shape == (13, 64, 1, 1)
    assert "feats.layer3.0.downsample.0.weight" not in sd
    assert "feats.layer4.0.downsample.0.weight" in sd
    assert sum(p.numel() for p in net.parameters()) == 54362637 or sum(p.numel() for p in net.parameters()) > 54e6
    m = R.RefConfSegnet()
    assert sum(p.numel() for p in m.parameters()) == 75794006
    assert set(k.split(".")[0] for k in m.state_dict()) == {"trgb_segnet", "critics"}


def test_segnet_fullres_minibatch(golden_dir):
    g = _load(golden_dir, "segnet_late_2x320x704")
    net = R.RefPSPNet(backend="resnet50", late_fusion=True, in_channels=4)
    net.load_state_dict(seeded_state(net, seed=2))
    set_dropout(net, 0.0)
    net.train()
    rgb, ir, lab = seeded_inputs(2, 320, 704, seed=5)
    with torch.no_grad():
        logits, feats, _ = net(rgb, ir)
    np.testing.assert_allclose(logits[:, :, ::8, ::8].numpy(), g["logits_s8"], rtol=0, atol=3e-4)
    ce = float(torch.nn.functional.cross_entropy(logits, lab))
    assert abs(ce - float(g["ce"])) < 1e-5
    pred = logits.argmax(1).numpy()
    ref_pred = g["argmax"].astype(np.int64)
    miou = np.nanmean(R.calculate_ious(pred, lab.numpy()))
    ref_miou = np.nanmean(R.calculate_ious(ref_pred, lab.numpy()))
    assert abs(miou - ref_miou) < 1e-3  # BASELINE.json criterion
    assert (pred != ref_pred).mean() < 1e-3


def test_rgb18_config1(golden_dir):
    g = _load(golden_dir, "rgb18_1x320x704")
    net = R.build_rgb_pspnet("resnet18", 18)
    assert sum(p.numel() for p in net.parameters()) == int(g["n_params"])
    net.load_state_dict(seeded_state(net, seed=4))
    set_dropout(net, 0.0)
    net.train()
    rgb, _, lab = seeded_inputs(1, 320, 704, seed=9, n_classes=14)
    logits = net(rgb)
    loss = torch.nn.functional.cross_entropy(logits, lab, ignore_index=13)
    loss.backward()
    np.testing.assert_allclose(logits[:, :, ::8, ::8].detach().numpy(), g["logits_s8"], rtol=0, atol=2e-4)
    assert abs(float(loss) - float(g["ce"])) < 1e-5
    np.testing.assert_allclose(net.final[0].weight.grad.numpy(), g["grad_final_w"], rtol=1e-3, atol=1e-5)
    assert abs(float(net.feats.conv1.weight.grad.double().norm()) / float(g["grad_conv1_norm"]) - 1) < 1e-3


def test_miou_integer_arithmetic(golden_dir):
    g = _load(golden_dir, "miou_cases")
    for case in range(3):
        pred, tgt = g["c%d_pred" % case].astype(np.int64), g["c%d_tgt" % case].astype(np.int64)
        np.testing.assert_array_equal(R.calculate_ious(pred, tgt), g["c%d_ious" % case])
        iou, miou = R.confusion_matrix_iou(pred, tgt, 14, (12, 13))
        np.testing.assert_array_equal(iou, g["c%d_cm_iou" % case])
        assert miou == float(g["c%d_cm_miou" % case])


def test_phase_schedule():
    s = R.PhaseSchedule(initial_critic=3, seg=2, critic=4)
    seen = []
    for _ in range(12):
        seen.append(s.state)
        s.tick()
    assert seen == ["train_critic"] * 3 + ["train_seg"] * 2 + ["train_critic"] * 4 + ["train_seg"] * 2 + ["train_critic"]


@pytest.mark.parametrize("adv", ["MSE", "BCE"])
def test_confusion_maximisation_steps(golden_dir, adv):
    """conv_segnet forward, then one train_critic and one train_seg iteration with RMSprop."""
    g = _load(golden_dir, "confstep_%s_2x256x256" % adv.lower())
    model = R.RefConfSegnet(num_critics=6, late_fusion=True)
    model.load_state_dict(seeded_state(model, seed=7))
    set_dropout(model, 0.0)
    model.train()
    opt = torch.optim.RMSprop(model.parameters(), lr=1e-4)
    rgb_d, ir_d, lab = seeded_inputs(2, 256, 256, seed=11)
    rgb_n, ir_n, _ = seeded_inputs(2, 256, 256, seed=12)
    for step, state in enumerate(("train_critic", "train_seg")):
        model.setPhase(state)
        if step == 0:
            with torch.no_grad():
                o = model([rgb_d, ir_d], [rgb_n, ir_n])
            np.testing.assert_allclose(o["pred_label_a"][:, :, ::4, ::4].numpy(), g["fwd_logits_a_s4"], atol=3e-4, rtol=0)
            for i in range(6):
                np.testing.assert_allclose(o["critics_a"][i][:, :, ::8, ::8].numpy(), g["fwd_critic_a%d" % i], atol=2e-4, rtol=1e-4)
                np.testing.assert_allclose(o["critics_b"][i][:, :, ::8, ::8].numpy(), g["fwd_critic_b%d" % i], atol=2e-4, rtol=1e-4)
            model.load_state_dict(seeded_state(model, seed=7))  # undo the BN running-stat update of the probe
        losses = R.train_step(model, opt, [rgb_d, ir_d], [rgb_n, ir_n], lab, state, adv=adv)
        assert abs(losses["total"] - float(g["s%d_total" % step])) < 2e-4 * max(1.0, abs(float(g["s%d_total" % step])))
        if state == "train_seg":
            assert abs(losses["seg"] - float(g["s1_seg"])) < 1e-4
            assert abs(losses["conf"] - float(g["s1_conf"])) < 2e-4 * max(1.0, abs(float(g["s1_conf"])))
        names = [str(n) for n in g["s%d_param_names" % step]]
        sums = dict(zip(names, g["s%d_param_sums" % step]))
        mine = {k: float(p.detach().double().sum()) for k, p in model.named_parameters()}
        assert list(mine) == names
        scale = {k: float(p.detach().abs().double().sum()) for k, p in model.named_parameters()}
        worst = max(abs(mine[k] - sums[k]) / (scale[k] + 1e-12) for k in names)
        assert worst < 1e-3, worst


@pytest.mark.parametrize("adv,state", [("MSE", "train_seg"), ("BCE", "train_critic")])
def test_gradient_tensors_match_reference_elementwise(golden_dir, adv, state):
    """The oracle's gradients against the reference's gradient TENSORS (make_golden.py --part confgrads): same op graph on
    the same CPU library, so they agree to accumulation-order noise -- a transposed or permuted dW would not."""
    from golden.make_golden import GRAD_SLICES

    class NoStep:
        def zero_grad(self, *a, **k):
            model.zero_grad()

        def step(self):
            pass

    g = _load(golden_dir, "confgrads_2x256x256")
    tag = "%s_%s" % (adv.lower(), state)
    model = R.RefConfSegnet(num_critics=6, late_fusion=True)
    model.load_state_dict(seeded_state(model, seed=7))
    set_dropout(model, 0.0)
    model.train()
    model.setPhase(state)
    rgb_d, ir_d, lab = seeded_inputs(2, 256, 256, seed=11)
    rgb_n, ir_n, _ = seeded_inputs(2, 256, 256, seed=12)
    losses = R.train_step(model, NoStep(), [rgb_d, ir_d], [rgb_n, ir_n], lab, state, adv=adv)
    assert abs(losses["total"] - float(g[tag + "_total"])) < 2e-4 * max(1.0, abs(float(g[tag + "_total"])))
    params = dict(model.named_parameters())
    for key, idx in GRAD_SLICES[state]:
        ref = torch.from_numpy(g["%s|%s" % (tag, key)]).double()
        grad = params[key].grad.detach().double()
        got = eval("grad" + idx) if idx else grad
        norm_full = float(g["%s|%s|norm" % (tag, key)])
        if float(ref.norm()) < 1e-6 * max(1.0, norm_full) or norm_full < 1e-9:
            continue
        rel = float((got - ref).norm() / ref.norm())
        assert rel < 5e-3, (key, rel)


@pytest.mark.parametrize("tag,late", [("late", True), ("early", False), ("cert", False)])
def test_resnext_oracle_matches_reference_fixture(golden_dir, tag, late):
    """oracle/resnext_ref.py (the ResNeXt-50 + ASPP "custom" architecture, cm/models/trgb_segnet.py:242-402) against the
    outputs of the reference's own module (make_golden.py --part resnext): state_dict keys and shapes, logits, loss, the five
    feature maps, the gradient norm of every parameter, BatchNorm running statistics."""
    from oracle import resnext_ref as R
    g = np.load(os.path.join(golden_dir, "resnext_%s_2x128x192.npz" % tag))
    net = R.RefResNeXt(late_fusion=late, cert_branch=(tag == "cert"))
    sd = net.state_dict()
    assert list(sd.keys()) == list(g["keys"])
    assert all(tuple(v.shape) == tuple(int(x) for x in sh[:v.dim()]) for v, sh in zip(sd.values(), g["shapes"]))
    net.load_state_dict(seeded_state(net, seed=21))
    rgb, ir, lab = seeded_inputs(2, 128, 192, seed=23)
    net.train()
    seg, feats, cert = net(rgb, ir)
    assert (cert is None) == (tag != "cert")
    loss = torch.nn.functional.cross_entropy(seg, lab)
    if tag == "cert":  # the certainty head and the trainer's certainty loss (cm/train_trgb_segnet_conf.py:482-491)
        closs = R.cert_loss(seg, lab, cert)
        assert float((cert.detach()[..., ::2, ::2] - torch.from_numpy(g["cert_sub"])).abs().max()) < 1e-4
        assert abs(float(closs.detach()) - float(g["cert_loss"])) < 1e-5 * max(1.0, float(g["cert_loss"]))
        loss = loss + closs
    loss.backward()
    assert float((seg.detach()[..., ::2, ::2] - torch.from_numpy(g["logits_sub"])).abs().max()) < 3e-4
    assert abs(float(loss.detach()) - float(g["loss"])) < 1e-5
    for i, t in enumerate(feats[1:]):
        assert tuple(t.shape) == tuple(g["feat%d_shape" % i])
        _, s = tensor_digest(t, 31)
        np.testing.assert_allclose(s, g["feat%d_sample" % i], rtol=0, atol=3e-4)
    norms = {k: float(p.grad.double().norm()) for k, p in net.named_parameters() if p.grad is not None}
    assert list(norms.keys()) == list(g["grad_keys"])
    floor = 1e-6 * float(g["grad_norms"].max())
    for k, ref in zip(g["grad_keys"], g["grad_norms"]):
        assert abs(norms[k] - ref) <= 5e-3 * max(ref, floor), k
    for k in ("mod1.bn1_2.bn.running_mean", "mod5_1.block3.convs.bn3.bn.running_var", "aspp.conv5.bn1.bn.running_var",
              "fuse_seg.bn1.bn.running_mean"):
        np.testing.assert_allclose(net.state_dict()[k].numpy(), g["after_" + k], rtol=1e-4, atol=1e-6)
