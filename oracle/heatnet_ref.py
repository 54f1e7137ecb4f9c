"""CPU oracle for the HeatNet hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement, in stock fp32 `torch` CPU ops, of the arithmetic the
reference performs on its training hot path.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import
this file (plus the diagnostic scripts under `tools/`, which use it the same
way: as the checker / the timed CPU side); the shipped package
(`heatnet-pub_amd/`) never does.

Pinning: the reference has no tests or golden vectors of its own (SURVEY.md
section 4), so this restatement is pinned by outputs of the reference's own
model code run in the build container: `tests/golden/make_golden.py` imports
`/root/reference/models/confusion_maximization/...` on CPU, copies one seeded
`state_dict` into both implementations and stores the reference's outputs as
`tests/golden/*.npz`; `tests/test_oracle_golden.py` checks this file against
them.  The arithmetic itself lives in a third-party dependency of the
reference (PyTorch, unpinned: its environment.yml is empty); torch 2.10 CPU is
the de-facto oracle arithmetic.

Every class cites the reference lines it follows; paths are relative to
/root/reference.  `state_dict()` keys equal the reference's (488 entries for
the late-fusion PSPNet).
"""
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# -----------------------------------------------------------------------------
# encoder: dilated ResNet, output stride 8
# -----------------------------------------------------------------------------


def _bn(c):
    return nn.BatchNorm2d(c)


class RefBottleneck(nn.Module):
    """models/confusion_maximization/models/extractors.py:66-102."""

    def __init__(self, cin, planes, stride, dilation, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _bn(planes)
        self.conv3 = nn.Conv2d(planes, 4 * planes, 1, bias=False)
        self.bn3 = _bn(4 * planes)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, 4 * planes, 1, stride=stride, bias=False), _bn(4 * planes))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


class RefBasicBlock(nn.Module):
    """models/extractors.py:34-63 (ResNet-18/34, BASELINE config 1 only)."""

    def __init__(self, cin, planes, stride, dilation, project):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = _bn(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = _bn(planes)
        self.downsample = None
        if project:
            self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride, bias=False), _bn(planes))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        skip = x if self.downsample is None else self.downsample(x)
        return F.relu(y + skip)


_DEPTHS = {
    "resnet18": (RefBasicBlock, 1, (2, 2, 2, 2)),
    "resnet34": (RefBasicBlock, 1, (3, 4, 6, 3)),
    "resnet50": (RefBottleneck, 4, (3, 4, 6, 3)),
    "resnet101": (RefBottleneck, 4, (3, 4, 23, 3)),
    "resnet152": (RefBottleneck, 4, (3, 8, 36, 3)),
}


class RefEncoder(nn.Module):
    """Dual-stream dilated ResNet.

    models/confusion_maximization/models/extractors.py:105-198.  layer3/layer4
    keep stride 1 and use dilation 2/4 in every block but the first
    (`_make_layer`, :156-170: block 0 is built with the default dilation 1).
    Late fusion adds conv1_2/bn1_2/layer1_2/layer2_2 and concatenates the two
    512-channel maps before layer3 (:192), so layer3.0 has no projection.
    """

    def __init__(self, backend="resnet50", late_fusion=False, in_channels=3):
        super().__init__()
        block, exp, depth = _DEPTHS[backend]
        self.late_fusion = late_fusion
        self.conv1 = nn.Conv2d(3 if late_fusion else in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = _bn(64)
        if late_fusion:
            self.conv1_2 = nn.Conv2d(1, 64, 7, stride=2, padding=3, bias=False)
            self.bn1_2 = _bn(64)

        def stage(cin, planes, n, stride, dilation):
            project = stride != 1 or cin != planes * exp
            blocks = [block(cin, planes, stride, 1, project)]
            blocks += [block(planes * exp, planes, 1, dilation, False) for _ in range(n - 1)]
            return nn.Sequential(*blocks)

        self.layer1 = stage(64, 64, depth[0], 1, 1)
        if late_fusion:
            self.layer1_2 = stage(64, 64, depth[0], 1, 1)
        self.layer2 = stage(64 * exp, 128, depth[1], 2, 1)
        if late_fusion:
            self.layer2_2 = stage(64 * exp, 128, depth[1], 2, 1)
        c3 = 128 * exp * (2 if late_fusion else 1)
        self.layer3 = stage(c3, 256, depth[2], 1, 2)
        self.layer4 = stage(256 * exp, 512, depth[3], 1, 4)
        # extractors.py:148-154
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    @staticmethod
    def _stem(conv, bn, x):
        return F.max_pool2d(F.relu(bn(conv(x))), 3, stride=2, padding=1)

    def forward(self, modal_1, modal_2=None):
        if not self.late_fusion and modal_2 is not None:
            modal_1 = torch.cat([modal_1, modal_2], dim=1)  # early fusion, extractors.py:173
        x1 = self._stem(self.conv1, self.bn1, modal_1)
        x2 = self.layer1(x1)
        x3 = self.layer2(x2)
        if self.late_fusion:
            t1 = self._stem(self.conv1_2, self.bn1_2, modal_2)
            t2 = self.layer1_2(t1)
            t3 = self.layer2_2(t2)
            x1, x2, x3 = torch.cat([x1, t1], 1), torch.cat([x2, t2], 1), torch.cat([x3, t3], 1)
        x4 = self.layer3(x3)
        x5 = self.layer4(x4)
        return [x5, x4, x3, x2, x1]


# -----------------------------------------------------------------------------
# PSP head and decoder
# -----------------------------------------------------------------------------


def _bilinear(x, size):
    # F.upsample(mode='bilinear') == interpolate(align_corners=False) (pspnet.py:23,39)
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)


class RefPSPModule(nn.Module):
    """models/confusion_maximization/models/pspnet.py:8-25."""

    def __init__(self, features, out_features=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList(
            [nn.Sequential(nn.AdaptiveAvgPool2d((s, s)), nn.Conv2d(features, features, 1, bias=False)) for s in sizes])
        self.bottleneck = nn.Conv2d(features * (len(sizes) + 1), out_features, 1)

    def forward(self, feats):
        hw = feats.shape[2:]
        pyramid = [_bilinear(stage(feats), hw) for stage in self.stages]
        return F.relu(self.bottleneck(torch.cat(pyramid + [feats], 1)))


class RefPSPUpsample(nn.Module):
    """models/confusion_maximization/models/pspnet.py:28-40."""

    def __init__(self, cin, cout):
        super().__init__()
        self.conv = nn.Sequential(nn.Conv2d(cin, cout, 3, padding=1), _bn(cout), nn.PReLU())

    def forward(self, x):
        return self.conv(_bilinear(x, (2 * x.shape[2], 2 * x.shape[3])))


class RefPSPNet(nn.Module):
    """HeatNet PSPNet: models/confusion_maximization/models/pspnet.py:43-76.

    `rgb_only=True` gives the top-level variant (models/pspnet.py:43-75):
    forward(x) -> logits, 18 classes by default.
    """

    def __init__(self, n_classes=13, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024,
                 backend="resnet50", late_fusion=False, in_channels=3, rgb_only=False):
        super().__init__()
        self.rgb_only = rgb_only
        self.feats = RefEncoder(backend, late_fusion, in_channels)
        self.psp = RefPSPModule(psp_size, 1024, sizes)
        self.drop_1 = nn.Dropout2d(p=0.3)
        self.up_1 = RefPSPUpsample(1024, 256)
        self.up_2 = RefPSPUpsample(256, 64)
        self.up_3 = RefPSPUpsample(64, 64)
        self.drop_2 = nn.Dropout2d(p=0.15)
        self.final = nn.Sequential(nn.Conv2d(64, n_classes, 1))

    def forward(self, modal_1, modal_2=None):
        f = self.feats(modal_1, modal_2)
        p = self.drop_1(self.psp(f[0]))
        p = self.drop_2(self.up_1(p))
        p = self.drop_2(self.up_2(p))
        p = self.drop_2(self.up_3(p))
        out = self.final(p)
        if self.rgb_only:
            return out
        return out, [out] + f, None


# -----------------------------------------------------------------------------
# critics and the confusion-maximisation wrapper
# -----------------------------------------------------------------------------


class RefFCDiscriminator(nn.Module):
    """models/confusion_maximization/discriminator_model.py:35-64."""

    def __init__(self, num_classes, ndf=64):
        super().__init__()
        self.conv1 = nn.Conv2d(num_classes, ndf, 4, stride=2, padding=1)
        self.conv2 = nn.Conv2d(ndf, 2 * ndf, 4, stride=2, padding=1)
        self.conv3 = nn.Conv2d(2 * ndf, 4 * ndf, 4, stride=2, padding=1)
        self.conv4 = nn.Conv2d(4 * ndf, 8 * ndf, 4, stride=2, padding=1)
        self.classifier = nn.Conv2d(8 * ndf, 1, 4, stride=2, padding=1)

    def forward(self, x):
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            x = F.leaky_relu(conv(x), 0.2)
        x = self.classifier(x)
        # nn.Upsample(scale_factor=32, mode='bilinear'): align_corners=None behaves as False
        return F.interpolate(x, scale_factor=32, mode="bilinear", align_corners=False)


def init_normal_(module):
    """models/confusion_maximization/utils.py:126-132 (`weights_init_normal`)."""
    name = module.__class__.__name__
    if "Conv" in name:
        nn.init.normal_(module.weight.data, 0.0, 0.02)
    elif "BatchNorm2d" in name:
        nn.init.normal_(module.weight.data, 1.0, 0.02)
        nn.init.constant_(module.bias.data, 0.0)


CRITIC_CHANNELS = {True: [13, 2048, 1024, 1024, 512, 128], False: [13, 2048, 1024, 512, 256, 64]}


class RefConfSegnet(nn.Module):
    """models/confusion_maximization/models/conf_segnet.py:22-140 (arch='pspnet', disc_arch='cyclegan')."""

    def __init__(self, num_critics=6, late_fusion=True, modalities="ir_rgb"):
        super().__init__()
        cin = (3 if "rgb" in modalities else 0) + (1 if "ir" in modalities else 0)
        self.trgb_segnet = RefPSPNet(backend="resnet50", late_fusion=late_fusion, in_channels=cin)
        self.trgb_segnet.apply(init_normal_)
        self.critics = nn.ModuleList([RefFCDiscriminator(c) for c in CRITIC_CHANNELS[late_fusion][:num_critics]])
        self.phase = "train_seg"

    def setPhase(self, phase):
        self.phase = phase
        seg = phase == "train_seg"
        for p in self.trgb_segnet.parameters():
            p.requires_grad = seg
        for p in self.critics.parameters():
            p.requires_grad = not seg

    def forward(self, input_a, input_b):
        pred_a, feat_a, _ = self.trgb_segnet(*input_a)
        pred_b, feat_b, _ = self.trgb_segnet(*input_b)
        out = {"critics_a": [c(feat_a[i]) for i, c in enumerate(self.critics)],
               "critics_b": [c(feat_b[i]) for i, c in enumerate(self.critics)],
               "pred_label_a": pred_a, "pred_label_b": pred_b, "cert_a": None, "cert_b": None,
               "inter_f_b": feat_b}
        return out


# -----------------------------------------------------------------------------
# losses and the training step (train_trgb_segnet_conf.py:237-244, 412-574)
# -----------------------------------------------------------------------------


def adv_criterion(kind):
    return nn.BCEWithLogitsLoss() if kind == "BCE" else nn.MSELoss()


def critic_loss(out, kind):
    """train_trgb_segnet_conf.py:437-446: day maps -> 1, night maps -> 0."""
    crit = adv_criterion(kind)
    la = sum(crit(c, torch.full_like(c, 1)) for c in out["critics_a"])
    lb = sum(crit(c, torch.full_like(c, 0)) for c in out["critics_b"])
    return la + lb


def confusion_loss(out, kind, weights=None, multidir=False):
    """train_trgb_segnet_conf.py:526-540: both domains pushed towards 'day' (1).

    The reference multiplies the (scalar, mean-reduced) criterion by a
    bilinear-resized all-ones map and takes the mean again, which is the
    criterion itself; that is what is restated here.
    """
    crit = adv_criterion(kind)
    n = len(out["critics_a"])
    weights = [1.0] * n if weights is None else weights
    total = 0.0
    for m, c in enumerate(out["critics_a"]):
        total = total + crit(c, torch.full_like(c, 0 if multidir else 1)) * weights[m]
    for m, c in enumerate(out["critics_b"]):
        total = total + crit(c, torch.full_like(c, 1)) * weights[m]
    return total


def train_step(model, optimizer, day, night, label_day, state, adv="MSE", conf_weight=0.1, night_teacher=None,
               teacher_modalities=("ir",), weight_ir_sup=False):
    """One iteration of train_trgb_segnet_conf.py:383-574 for `state` in {train_seg, train_critic}.

    Returns a dict of the scalar losses.  `optimizer.zero_grad()` leaves grads
    None (torch >= 2.0), so frozen-phase parameters are skipped by the optimizer.
    `night_teacher`: the night-supervision model of :209-229 (a frozen segmentation net fed the night IR image, or RGB+IR);
    its arg-max prediction is the label of the night pass in train_seg (:430-434, 455-457).  `weight_ir_sup`: the
    `--weight_ir_sup` variant without a cert branch (:467-477, 531-540), line for line: certainty = max over classes of
    softmax applied to the teacher's softmax output; night CE per pixel times certainty, averaged; each confusion term
    times the (1 - certainty) map resized bilinearly to the critic's output size (batch used as the channel axis).
    """
    optimizer.zero_grad()
    out = model(day, night)
    losses = {}
    if state == "train_seg":
        seg = F.cross_entropy(out["pred_label_a"], label_day)
        if night_teacher is not None:
            with torch.no_grad():
                t_in = [night[0], night[1]] if "rgb" in teacher_modalities else [night[1]]
                probs = F.softmax(night_teacher(*t_in)[0], dim=1)
                pseudo = torch.argmax(probs, 1)
            conf_maps = None
            if weight_ir_sup:
                per_pixel = F.cross_entropy(out["pred_label_b"], pseudo, reduction="none")
                cert = F.softmax(probs, dim=1).max(1)[0]
                seg_night = torch.mean(cert * per_pixel)
                conf_maps = (torch.full_like(cert, 1.0) - cert).unsqueeze(0)
            else:
                seg_night = F.cross_entropy(out["pred_label_b"], pseudo)
            losses["seg_night"] = float(seg_night.detach())
            seg = seg + seg_night
        if night_teacher is not None and weight_ir_sup:
            crit = adv_criterion(adv)
            conf = 0.0
            for c in list(out["critics_a"]) + list(out["critics_b"]):
                wmap = F.interpolate(conf_maps, size=(c.size(2), c.size(3)), mode="bilinear")
                conf = conf + torch.mean(wmap * crit(c, torch.full_like(c, 1))) * 1.0
        else:
            conf = confusion_loss(out, adv)
        total = seg + conf_weight * conf
        losses.update(seg=float(seg.detach()), conf=float(conf.detach()))
    else:
        total = critic_loss(out, adv)
    total.backward()
    optimizer.step()
    losses["total"] = float(total.detach())
    return losses


class PhaseSchedule:
    """train_trgb_segnet_conf.py:361-367,577-592: critic warm-up, then 50 seg / 500 critic iterations."""

    def __init__(self, initial_critic=1000, seg=50, critic=500):
        self.state, self.counter = "train_critic", initial_critic
        self.seg, self.critic = seg, critic

    def tick(self):
        """Call once per iteration after the step; returns True when the phase flips."""
        self.counter -= 1
        if self.counter != 0:
            return False
        if self.state == "train_seg":
            self.state, self.counter = "train_critic", self.critic
        else:
            self.state, self.counter = "train_seg", self.seg
        return True


# -----------------------------------------------------------------------------
# mIoU (integer arithmetic)
# -----------------------------------------------------------------------------


def calculate_ious(pred, target, n_classes=13):
    """models/confusion_maximization/utils.py:134-163: class 12 (background) and 13 (ignore) skipped."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    target = np.asarray(target).reshape(-1).astype(np.int64)
    keep = target != 13
    ious = []
    for cls in range(n_classes):
        if cls in (12, 13):
            continue
        p, t = pred == cls, target == cls
        inter = int(np.count_nonzero(p & t))
        union = int(np.count_nonzero(p & keep)) + int(np.count_nonzero(t & keep)) - inter
        ious.append(float("nan") if union == 0 else inter / max(union, 1))
    return np.array(ious)


def confusion_matrix_iou(pred, target, num_classes, ignore_index=()):
    """scripts/iou_eval.py:19-182 (`IoU.add` + `IoU.value`): rows = target, columns = prediction."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    target = np.asarray(target).reshape(-1).astype(np.int64)
    conf = np.bincount(pred + num_classes * target, minlength=num_classes ** 2).reshape(num_classes, num_classes)
    conf = conf.astype(np.int64)
    for idx in ignore_index:
        conf[:, idx] = 0
        conf[idx, :] = 0
    tp = np.diag(conf)
    fp = conf.sum(0) - tp
    fn = conf.sum(1) - tp
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (tp + fp + fn)
    return iou, np.nanmean(iou)


def build_rgb_pspnet(backend="resnet18", n_classes=18):
    """models/build_net.py:7-15 table (pretrained weights never fetched here)."""
    psp_size, deep = (512, 256) if backend in ("resnet18", "resnet34") else (2048, 1024)
    return RefPSPNet(n_classes=n_classes, psp_size=psp_size, deep_features_size=deep, backend=backend, rgb_only=True)
