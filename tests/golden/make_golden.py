"""Generate tests/golden/*.npz by running the REFERENCE's model code on CPU.

Runs only in the build container (it reads /root/reference); the fixtures it
writes are what travels.  Usage:

    python tests/golden/make_golden.py            # all parts, one subprocess each
    python tests/golden/make_golden.py --part segnet

Parts run in separate processes because the reference has two different
packages both called `models`.  torchvision is not
installed; the reference imports four names from it at module top that the
ResNet path never calls, so placeholder modules that raise when called are put
in sys.modules, and `nn.Module.cuda` is made a no-op (no GPU here).
"""
import argparse
import os
import subprocess
import sys
import types

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _seeded import seeded_inputs, seeded_state, set_dropout, tensor_digest  # noqa: E402

REF = "/root/reference"
CM = REF + "/models/confusion_maximization"


def _shim_environment():
    def _unavailable(*a, **k):
        raise RuntimeError("torchvision is not installed; this name must not be called")
    names = {"torchvision": {}, "torchvision.models": {},
             "torchvision.models.densenet": {"densenet121": _unavailable, "densenet161": _unavailable},
             "torchvision.models.squeezenet": {"squeezenet1_1": _unavailable},
             "torchvision.models.resnet": {"load_state_dict_from_url": _unavailable}}
    for name, attrs in names.items():
        mod = types.ModuleType(name)
        mod.__dict__.update(attrs)
        sys.modules[name] = mod
    nn.Module.cuda = lambda self, *a, **k: self
    torch.set_num_threads(8)


def _save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote", path, os.path.getsize(path) // 1024, "KiB")


def _digests(prefix, tensors, stride, out):
    for i, t in enumerate(tensors):
        d, s = tensor_digest(t, stride)
        out["%s%d_digest" % (prefix, i)] = d
        out["%s%d_sample" % (prefix, i)] = s
        out["%s%d_shape" % (prefix, i)] = np.array(t.shape)


def part_segnet():
    """Late- and early-fusion HeatNet PSPNet forward (conf-max models/pspnet.py:60-76)."""
    sys.path.insert(0, CM)
    from models.pspnet import PSPNet
    for tag, late, cin in (("late", True, 4), ("early", False, 4)):
        net = PSPNet(sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet50",
                     in_channels=cin, pretrained=False, late_fusion=late)
        net.load_state_dict(seeded_state(net, seed=1))
        set_dropout(net, 0.0)
        rgb, ir, _ = seeded_inputs(2, 64, 96, seed=3)
        out = {}
        net.train()
        logits, feats, _ = net(rgb, ir)
        out["train_logits"] = logits.detach().numpy()
        _digests("train_feat", feats[1:], 97, out)
        sd = net.state_dict()
        for k in ("feats.bn1.running_mean", "feats.bn1.running_var", "feats.layer4.2.bn3.running_mean",
                  "feats.layer4.2.bn3.running_var", "up_3.conv.1.running_var"):
            out["after_" + k] = sd[k].numpy().copy()
        out["after_nbt"] = np.array(int(sd["feats.bn1.num_batches_tracked"]))
        net.load_state_dict(seeded_state(net, seed=1))
        net.eval()
        with torch.no_grad():
            logits, feats, _ = net(rgb, ir)
        out["eval_logits"] = logits.numpy()
        _digests("eval_feat", feats[1:], 97, out)
        _save("segnet_%s_2x64x96" % tag, **out)

    # BASELINE config-2 shape, fixed mini-batch for the mIoU criterion
    net = PSPNet(sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet50",
                 in_channels=4, pretrained=False, late_fusion=True)
    net.load_state_dict(seeded_state(net, seed=2))
    set_dropout(net, 0.0)
    net.train()
    rgb, ir, lab = seeded_inputs(2, 320, 704, seed=5)
    with torch.no_grad():
        logits, feats, _ = net(rgb, ir)
    loss = torch.nn.CrossEntropyLoss()(logits, lab)
    out = {"logits_s8": logits[:, :, ::8, ::8].numpy().copy(), "argmax": logits.argmax(1).numpy().astype(np.uint8),
           "ce": np.array(float(loss))}
    _digests("feat", feats, 1009, out)
    _save("segnet_late_2x320x704", **out)


def part_confstep():
    """conv_segnet forward + one train_critic and one train_seg iteration (train_trgb_segnet_conf.py:412-574)."""
    sys.path.insert(0, CM)
    from models import conf_segnet
    for adv in ("MSE", "BCE"):
        model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet",
                                        late_fusion=True)
        model.load_state_dict(seeded_state(model, seed=7))
        set_dropout(model, 0.0)
        model.train()
        crit = torch.nn.BCEWithLogitsLoss() if adv == "BCE" else torch.nn.MSELoss()
        ce = torch.nn.CrossEntropyLoss()
        opt = torch.optim.RMSprop(model.parameters(), lr=1e-4)
        rgb_d, ir_d, lab = seeded_inputs(2, 256, 256, seed=11)
        rgb_n, ir_n, _ = seeded_inputs(2, 256, 256, seed=12)
        out = {}
        for step, state in enumerate(("train_critic", "train_seg")):
            model.setPhase(state)
            opt.zero_grad()
            o = model([rgb_d, ir_d], [rgb_n, ir_n])
            # lines 437-446
            ta = sum(torch.sum(crit(c, torch.full_like(c, 1))) for c in o["critics_a"])
            tb = sum(torch.sum(crit(c, torch.full_like(c, 0))) for c in o["critics_b"])
            if state == "train_seg":
                seg = ce(o["pred_label_a"], lab)  # line 452
                ones = torch.ones_like(o["critics_a"][0])
                conf = torch.zeros(1)
                for c in list(o["critics_a"]) + list(o["critics_b"]):  # lines 536-540, weights all 1
                    wmap = torch.nn.functional.interpolate(ones, size=(c.size(2), c.size(3)), mode="bilinear")
                    conf = conf + torch.mean(wmap * crit(c, torch.full_like(c, 1))) * 1.0
                total = seg + 0.1 * conf
                out["s%d_seg" % step] = np.array(float(seg))
                out["s%d_conf" % step] = np.array(float(conf))
            else:
                total = ta + tb
            out["s%d_total" % step] = np.array(float(total))
            out["s%d_critic_total" % step] = np.array(float(ta + tb))
            if step == 0:
                out["fwd_logits_a_s4"] = o["pred_label_a"][:, :, ::4, ::4].detach().numpy().copy()
                for i, c in enumerate(o["critics_a"]):
                    out["fwd_critic_a%d" % i] = c.detach().numpy()[:, :, ::8, ::8].copy()
                    out["fwd_critic_b%d" % i] = o["critics_b"][i].detach().numpy()[:, :, ::8, ::8].copy()
            total.backward()
            names, norms = [], []
            for k, p in model.named_parameters():
                if p.grad is not None:
                    names.append(k)
                    norms.append(float(p.grad.double().norm()))
            out["s%d_grad_names" % step] = np.array(names)
            out["s%d_grad_norms" % step] = np.array(norms)
            opt.step()
            names, sums = [], []
            for k, p in model.named_parameters():
                names.append(k)
                sums.append(float(p.detach().double().sum()))
            out["s%d_param_names" % step] = np.array(names)
            out["s%d_param_sums" % step] = np.array(sums)
        _save("confstep_%s_2x256x256" % adv.lower(), **out)


# Parameters whose gradient TENSORS (not norms) are stored by part_confgrads: (key, index expression or None = whole).
# A norm or a sum is invariant to a permuted / transposed dW; these slices are not.
GRAD_SLICES = {
    "train_seg": [
        ("trgb_segnet.feats.conv1.weight", None), ("trgb_segnet.feats.conv1_2.weight", None),
        ("trgb_segnet.feats.bn1_2.weight", None), ("trgb_segnet.feats.layer1.0.conv1.weight", None),
        ("trgb_segnet.feats.layer1_2.2.conv3.weight", "[:32, :32]"), ("trgb_segnet.feats.layer2.0.conv2.weight", "[:16, :32]"),
        ("trgb_segnet.feats.layer2.0.downsample.0.weight", "[:32, :64]"),
        ("trgb_segnet.feats.layer2.1.bn2.weight", None), ("trgb_segnet.feats.layer2.1.bn2.bias", None),
        ("trgb_segnet.feats.layer3.2.conv2.weight", "[:16, :16]"), ("trgb_segnet.feats.layer3.0.conv1.weight", "[:16, ::8]"),
        ("trgb_segnet.feats.layer4.1.conv2.weight", "[:8, :32]"), ("trgb_segnet.feats.layer4.2.conv3.weight", "[::64, ::16]"),
        ("trgb_segnet.psp.stages.2.1.weight", "[:32, :32]"), ("trgb_segnet.psp.bottleneck.weight", "[:16, ::64]"),
        ("trgb_segnet.psp.bottleneck.bias", None), ("trgb_segnet.up_1.conv.0.weight", "[:16, :64]"),
        ("trgb_segnet.up_1.conv.1.weight", None), ("trgb_segnet.up_1.conv.2.weight", None),
        ("trgb_segnet.up_3.conv.0.weight", "[:, :16]"), ("trgb_segnet.final.0.weight", None),
        ("trgb_segnet.final.0.bias", None),
    ],
    "train_critic": [
        ("critics.0.conv1.weight", None), ("critics.0.conv1.bias", None), ("critics.1.conv2.weight", "[:8, :64]"),
        ("critics.2.conv3.bias", None), ("critics.3.conv4.weight", "[:4, :64]"), ("critics.4.conv1.weight", "[:8, ::8]"),
        ("critics.5.classifier.weight", None), ("critics.5.classifier.bias", None),
    ],
}


def part_confgrads():
    """Elementwise gradient tensors of ONE train_seg and ONE train_critic iteration, each from the seeded state
    (train_trgb_segnet_conf.py:412-574), for both adversarial losses."""
    sys.path.insert(0, CM)
    from models import conf_segnet
    out = {}
    for adv in ("MSE", "BCE"):
        crit = torch.nn.BCEWithLogitsLoss() if adv == "BCE" else torch.nn.MSELoss()
        ce = torch.nn.CrossEntropyLoss()
        rgb_d, ir_d, lab = seeded_inputs(2, 256, 256, seed=11)
        rgb_n, ir_n, _ = seeded_inputs(2, 256, 256, seed=12)
        for state in ("train_seg", "train_critic"):
            model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet",
                                            late_fusion=True)
            model.load_state_dict(seeded_state(model, seed=7))
            set_dropout(model, 0.0)
            model.train()
            model.setPhase(state)
            o = model([rgb_d, ir_d], [rgb_n, ir_n])
            if state == "train_seg":
                seg = ce(o["pred_label_a"], lab)
                ones = torch.ones_like(o["critics_a"][0])
                conf = torch.zeros(1)
                for c in list(o["critics_a"]) + list(o["critics_b"]):
                    wmap = torch.nn.functional.interpolate(ones, size=(c.size(2), c.size(3)), mode="bilinear")
                    conf = conf + torch.mean(wmap * crit(c, torch.full_like(c, 1))) * 1.0
                total = seg + 0.1 * conf
            else:
                total = sum(torch.sum(crit(c, torch.full_like(c, 1))) for c in o["critics_a"]) + \
                    sum(torch.sum(crit(c, torch.full_like(c, 0))) for c in o["critics_b"])
            total.backward()
            params = dict(model.named_parameters())
            tag = "%s_%s" % (adv.lower(), state)
            out[tag + "_total"] = np.array(float(total))
            for key, idx in GRAD_SLICES[state]:
                g = params[key].grad
                out["%s|%s|norm" % (tag, key)] = np.array(float(g.double().norm()))
                out["%s|%s" % (tag, key)] = (eval("g" + idx) if idx else g).numpy().copy()
    _save("confgrads_2x256x256", **out)


def part_feedback():
    """`--feedback_seg`: layout of conv_segnet with DownNets (conf_segnet.py:61-70) and one DownNet forward / backward
    (downscale_network.py:23-35) on a 12-channel map."""
    sys.path.insert(0, CM)
    from models import conf_segnet
    from models.downscale_network import DownNet
    model = conf_segnet.conv_segnet(pretrained=False, disc_arch="cyclegan", num_critics=6, arch="pspnet",
                                    late_fusion=True, feedback_seg=True)
    keys = [k for k in model.state_dict().keys() if not k.startswith("trgb_segnet.")]
    out = {"keys": np.array(keys), "shapes": np.array([list(model.state_dict()[k].shape) + [0] * (4 - model.state_dict()[k].dim())
                                                       for k in keys])}
    try:  # the reference itself cannot run this configuration on the 13-class PSPNet head
        x = [torch.randn(1, 3, 256, 256), torch.randn(1, 1, 256, 256)]
        model(x, x)
        out["forward_error"] = np.array("")
    except RuntimeError as e:
        out["forward_error"] = np.array(str(e)[:200])
    net = DownNet(3)
    net.load_state_dict(seeded_state(net, seed=13))
    r = np.random.RandomState(5)
    x = torch.from_numpy(r.standard_normal((2, 12, 32, 48)).astype(np.float32)).requires_grad_()
    gy = torch.from_numpy(r.standard_normal((2, 12, 4, 6)).astype(np.float32))
    y = net(x)
    y.backward(gy)
    out.update(x=x.detach().numpy(), gy=gy.numpy(), y=y.detach().numpy(), dx=x.grad.numpy())
    for k, p in net.named_parameters():
        out["dw|" + k] = p.grad.numpy()
    _save("feedback_seg", **out)


def part_rgb18():
    """BASELINE config 1: top-level PSPNet, ResNet-18, 1x3x320x704, forward + CE(ignore_index=13)."""
    sys.path.insert(0, REF)
    from models.pspnet import PSPNet  # top-level copy (models/pspnet.py:43-75)
    net = PSPNet(sizes=(1, 2, 3, 6), psp_size=512, deep_features_size=256, backend="resnet18", pretrained=False)
    net.load_state_dict(seeded_state(net, seed=4))
    set_dropout(net, 0.0)
    net.train()
    rgb, _, lab = seeded_inputs(1, 320, 704, seed=9, n_classes=14, ignore_frac=0.0)  # labels 0..13, 13 = ignore
    logits = net(rgb)
    loss = torch.nn.CrossEntropyLoss(ignore_index=13)(logits, lab)
    loss.backward()
    out = {"logits_s8": logits[:, :, ::8, ::8].detach().numpy().copy(), "ce": np.array(float(loss)),
           "n_params": np.array(sum(p.numel() for p in net.parameters())),
           "grad_final_w": net.final[0].weight.grad.numpy().copy(),
           "grad_conv1_norm": np.array(float(net.feats.conv1.weight.grad.double().norm()))}
    _save("rgb18_1x320x704", **out)


def part_miou():
    """Integer arithmetic: utils.calculate_ious (utils.py:134-163) and scripts/iou_eval.IoU."""
    sys.path.insert(0, CM)
    sys.path.insert(0, REF + "/scripts")
    import utils as ref_utils
    from iou_eval import IoU
    r = np.random.RandomState(77)
    out = {}
    for case, (n, with_ignore) in enumerate(((3, False), (2, True), (1, True))):
        pred = r.randint(0, 13, (n, 40, 56))
        tgt = r.randint(0, 14 if with_ignore else 13, (n, 40, 56))
        if case == 2:  # classes 3 and 5 absent everywhere -> NaN entries
            pred[(pred == 3) | (pred == 5)] = 0
            tgt[(tgt == 3) | (tgt == 5)] = 0
        out["c%d_pred" % case] = pred.astype(np.uint8)
        out["c%d_tgt" % case] = tgt.astype(np.uint8)
        out["c%d_ious" % case] = ref_utils.calculate_ious(torch.from_numpy(pred).float(), torch.from_numpy(tgt).float())
        m = IoU(14, False, [12, 13])  # scripts/main.py:validate
        m.add(torch.from_numpy(pred), torch.from_numpy(tgt))
        iou, miou = m.value()
        out["c%d_cm_iou" % case] = iou
        out["c%d_cm_miou" % case] = np.array(miou)
    _save("miou_cases", **out)


RESNEXT_GRAD_KEYS = ["mod1.conv1_2.weight", "mod1_2.conv3.weight", "mod2_1.block1.convs.conv2.weight", "mod2_1.block1.proj_conv.weight",
                     "mod2_2.block3.convs.conv1.weight", "mod3_1.block1.convs.conv1.weight", "mod3_1.block2.convs.conv2.weight",
                     "mod4_1.block3.convs.conv2.weight", "mod4_1.block6.convs.conv3.weight", "mod5_1.block1.proj_conv.weight",
                     "mod5_1.block2.convs.conv2.weight", "mod5_1.block3.bn1.bn.weight", "bn_out_1.bn.bias",
                     "aspp.conv1.conv1.weight", "aspp.conv3.conv1.weight", "aspp.conv4.bn1.bn.weight", "aspp.conv5.conv1.weight",
                     "aspp.fuse_conv.conv1.weight", "aspp.final_conv.conv1.weight", "up_seg_2.weight", "fuse_seg.conv1.weight",
                     "fuse_seg.bn2.bn.weight", "fuse_seg.conv3.weight", "fuse_seg.up.weight",
                     "aspp.final_conv2.conv1.weight", "up_seg_2_cert.weight", "fuse_seg_cert.conv1.weight", "fuse_seg_cert.bn2.bn.weight",
                     "fuse_seg_cert.conv3.weight", "fuse_seg_cert.up.weight"]


def part_resnext():
    """The ResNeXt-50 (groups = 64) + ASPP "custom" architecture (conf-max models/trgb_segnet.py:242-402), late fusion, as
    conf_segnet builds it (models/conf_segnet.py:38-43): train-mode forward, cross-entropy, backward."""
    sys.path.insert(0, CM)
    from models import trgb_segnet
    for tag, late in (("late", True), ("early", False), ("cert", False)):
        net = trgb_segnet.ResNeXt(structure=[3, 4, 6, 3], input_channels=4, cert_branch=(tag == "cert"), late_fusion=late)
        net.load_state_dict(seeded_state(net, seed=21))
        rgb, ir, lab = seeded_inputs(2, 128, 192, seed=23)
        net.train()
        seg, feats, cert = net(rgb, ir)
        assert (cert is None) == (tag != "cert")
        loss = torch.nn.functional.cross_entropy(seg, lab)
        extra = {}
        if tag == "cert":
            # the trainer's certainty loss, cm/train_trgb_segnet_conf.py:482-491 (the script itself cannot be imported: wandb,
            # datasets); utils.make_one_hot scatters into a torch.cuda.FloatTensor -- the same one-hot built on the CPU
            one_hot = torch.zeros(lab.size(0), seg.size(1), lab.size(1), lab.size(2)).scatter_(1, lab.unsqueeze(1), 1)
            c = torch.sum((one_hot.float() * torch.nn.functional.softmax(seg, dim=1)), 1)
            c = torch.ones_like(c) - c
            cert_loss = torch.mean((cert - c) ** 2) * 10
            extra = {"cert_sub": cert.detach()[..., ::2, ::2].numpy().copy(), "cert_loss": np.array(float(cert_loss.detach()))}
            loss = loss + cert_loss
        loss.backward()
        d, _ = tensor_digest(seg, 1)
        out = {"logits_sub": seg.detach()[..., ::2, ::2].numpy().copy(), "logits_digest": d, "loss": np.array(float(loss.detach())),
               **extra,
               "keys": np.array(list(net.state_dict().keys())),
               "shapes": np.array([list(v.shape) + [0] * (4 - v.dim()) for v in net.state_dict().values()])}
        _digests("feat", feats[1:], 31, out)
        params = dict(net.named_parameters())
        out["grad_keys"] = np.array([k for k, p_ in params.items() if p_.grad is not None])
        out["grad_norms"] = np.array([float(p_.grad.double().norm()) for p_ in params.values() if p_.grad is not None])
        for k in RESNEXT_GRAD_KEYS:
            if late or "_2." not in k.split(".")[0] + ".":
                if k in params and params[k].grad is not None:
                    g = params[k].grad
                    out["grad|" + k] = g.reshape(-1)[:: max(1, g.numel() // 4096)].numpy().copy()
        sd = net.state_dict()
        for k in ("mod1.bn1_2.bn.running_mean", "mod5_1.block3.convs.bn3.bn.running_var", "aspp.conv5.bn1.bn.running_var",
                  "fuse_seg.bn1.bn.running_mean"):
            out["after_" + k] = sd[k].numpy()
        _save("resnext_%s_2x128x192" % tag, **out)


PARTS = {"resnext": part_resnext, "segnet": part_segnet, "confstep": part_confstep, "confgrads": part_confgrads, "feedback": part_feedback,
         "rgb18": part_rgb18,
         "miou": part_miou}

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", choices=sorted(PARTS))
    args = ap.parse_args()
    if args.part is None:
        for p in PARTS:
            subprocess.check_call([sys.executable, os.path.abspath(__file__), "--part", p])
    else:
        _shim_environment()
        torch.manual_seed(0)
        PARTS[args.part]()
