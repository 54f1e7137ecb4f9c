"""CPU oracle (TEST INFRASTRUCTURE -- only tests/, smoke() and bench.py's cpu_baseline may import it): a restatement, in stock
fp32 torch ops, of the reference's ResNeXt-50 + ASPP "custom" segmentation network
(models/confusion_maximization/models/trgb_segnet.py), the default `arch` of `conv_segnet` (conf_segnet.py:23,38-43).

Pinned by outputs of the reference itself: tests/golden/make_golden.py --part resnext imports the reference's module on the CPU
and stores logits, feature digests, gradient norms / slices and BatchNorm statistics (resnext_{late,early}_2x128x192.npz);
tests/test_oracle_golden.py checks this file against them.  Same `state_dict` keys as the reference.

What the network is (file:line of the reference):
  * `init_block` (:230-240): conv3x3 s2 -> BN+ReLU -> conv3x3 -> BN+ReLU -> conv3x3 -> max-pool 3x3 s2; one per modality when
    `late_fusion` (:288-293).
  * four modules of PRE-ACTIVATION bottlenecks (`PreActBottleneck`, :47-129): BN+ReLU of the block input, then conv1x1 (stride)
    -> BN+ReLU -> conv3x3 with 64 groups (dilation) -> BN+ReLU -> conv1x1, plus the input (or a 1x1 projection of the
    normalised input where shape changes).  Channels (128, 128, 256) doubling per module; strides / dilations from
    `_stride_dilation` (:404-417) with dilation = [1, 1, 2, 4]: module 3 starts with stride 2, modules 4 and 5 are dilated.
    Late fusion: module 2 exists twice (RGB, thermal), module 3 takes their concatenation (:299-318, 356-366).
  * `bn_out_1` (BN+ReLU) -> ASPP (:133-183): 1x1 and three dilated 3x3 branches (12, 24, 36) -> 256 each, a global-average
    branch, 1x1 fusion to 256, 1x1 to the 13 classes.
  * decoder (:376-381): ConvTranspose2d(13, 13, 4, 2, 1), concatenation with the module-2 features, three 3x3 convolutions
    (`fuseModule`, :201-209) and ConvTranspose2d(13, 13, 8, 4, 2) to the input resolution.
  * outputs (:391-393): logits, [logits, cat(fusion, pooled module-2 features), out_4, out_3, out_2, out_1], None.
"""
from collections import OrderedDict

import torch
import torch.nn as nn
import torch.nn.functional as F


def abn(c):
    return nn.Sequential(OrderedDict([("bn", nn.BatchNorm2d(c)), ("act", nn.ReLU(inplace=True))]))


class PreActBottleneck(nn.Module):
    def __init__(self, cin, channels, stride=1, dilation=1, groups=64):
        super().__init__()
        c1, c2, c3 = channels
        self.bn1 = abn(cin)
        self.convs = nn.Sequential(OrderedDict([
            ("conv1", nn.Conv2d(cin, c1, 1, stride=stride, bias=False)),
            ("bn2", abn(c1)),
            ("conv2", nn.Conv2d(c1, c2, 3, padding=dilation, dilation=dilation, groups=groups, bias=False)),
            ("bn3", abn(c2)),
            ("conv3", nn.Conv2d(c2, c3, 1, bias=False))]))
        if stride != 1 or cin != c3:
            self.proj_conv = nn.Conv2d(cin, c3, 1, stride=stride, bias=False)

    def forward(self, x):
        a = self.bn1(x)
        shortcut = self.proj_conv(a) if hasattr(self, "proj_conv") else x
        return self.convs(a) + shortcut


def conv_abn(cin, cout, k, dil):
    pad = dil if dil != 1 else (1 if k == 3 else 0)
    return nn.Sequential(OrderedDict([("conv1", nn.Conv2d(cin, cout, k, padding=pad, dilation=dil, bias=False)),
                                      ("bn1", abn(cout))]))


class RefASPP(nn.Module):
    def __init__(self, cin, classes, rates=(1, 12, 24, 36), kernels=(1, 3, 3, 3)):
        super().__init__()
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), conv_abn(cin, 256, kernels[i], rates[i]))
        self.conv5 = conv_abn(cin, 256, 1, 1)
        self.fuse_conv = conv_abn(256 * 5, 256, 1, 1)
        self.final_conv = conv_abn(256, classes, 1, 1)
        self.final_conv2 = conv_abn(256, 1, 1, 1)

    def forward(self, x):
        xs = [getattr(self, "conv%d" % i)(x) for i in range(1, 5)]
        g = self.conv5(x.mean(dim=(2, 3), keepdim=True))
        xs.append(g.expand(-1, -1, x.size(2), x.size(3)))  # bilinear resampling of a 1 x 1 map is a broadcast
        fusion = self.fuse_conv(torch.cat(xs, 1))
        return self.final_conv(fusion), fusion, self.final_conv2(fusion)


def init_block(cin):
    return nn.Sequential(OrderedDict([
        ("conv1_2", nn.Conv2d(cin, 64, 3, stride=2, padding=1, bias=False)), ("bn1_2", abn(64)),
        ("conv2", nn.Conv2d(64, 64, 3, padding=1, bias=False)), ("bn2", abn(64)),
        ("conv3", nn.Conv2d(64, 64, 3, padding=1, bias=False)), ("pool", nn.MaxPool2d(3, stride=2, padding=1))]))


def fuse_module(cin, cout):
    return nn.Sequential(OrderedDict([
        ("conv1", nn.Conv2d(cin, cin, 3, padding=1, bias=False)), ("bn1", abn(cin)),
        ("conv2", nn.Conv2d(cin, cin, 3, padding=1, bias=False)), ("bn2", abn(cin)),
        ("conv3", nn.Conv2d(cin, cout, 3, padding=1, bias=False)),
        ("up", nn.ConvTranspose2d(cout, cout, 8, 4, 2, bias=False))]))


class RefResNeXt(nn.Module):
    def __init__(self, structure=(3, 4, 6, 3), groups=64, classes=13, dilation=(1, 1, 2, 4), base_channels=(128, 128, 256),
                 input_channels=4, late_fusion=False, cert_branch=False):
        super().__init__()
        self.late_fusion = late_fusion
        self.cert_branch = cert_branch
        if late_fusion:
            self.mod1, self.mod1_2 = init_block(3), init_block(1)
        else:
            self.mod1 = init_block(input_channels)
        cin, channels = 64, list(base_channels)
        for mod_id, num in enumerate(structure):
            start = cin
            for rep in range(2 if (late_fusion and mod_id == 0) else 1):
                cin = start * 2 if (mod_id == 1 and late_fusion) else start
                blocks = []
                for b in range(num):
                    if dilation[mod_id] == 1:
                        s, d = (2 if (mod_id > 0 and b == 0) else 1), 1
                    else:
                        s, d = 1, dilation[mod_id]
                    blocks.append(("block%d" % (b + 1), PreActBottleneck(cin, channels, s, d, groups)))
                    cin = channels[-1]
                self.add_module("mod%d_%d" % (mod_id + 2, rep + 1), nn.Sequential(OrderedDict(blocks)))
            channels = [c * 2 for c in channels]
        self.bn_out_1, self.bn_out_2 = abn(cin), abn(cin)  # (bn_out_2 is never used by forward; it is in the state_dict)
        self.aspp = RefASPP(cin, classes)
        self.up_seg_2 = nn.ConvTranspose2d(classes, classes, 4, 2, 1, bias=False)
        self.up_borders_3 = nn.ConvTranspose2d(5, 5, 4, 2, 1, bias=False)  # unused heads of the reference, kept for the keys
        self.up_inst_2 = nn.ConvTranspose2d(5, 5, 4, 2, 1, bias=False)
        self.fuse_seg = fuse_module((512 if late_fusion else 256) + classes, classes)
        if cert_branch:  # cm/models/trgb_segnet.py:344-346
            self.up_seg_2_cert = nn.ConvTranspose2d(1, 1, 4, 2, 1, bias=False)
            self.fuse_seg_cert = fuse_module(256 + 1, 1)

    def forward(self, modal_1, modal_2=None):
        if self.late_fusion and modal_2 is not None:
            o1, o1b = self.mod1(modal_1), self.mod1_2(modal_2)
            o2, o2b = self.mod2_1(o1), self.mod2_2(o1b)
            out_1, out_2 = torch.cat([o1, o1b], 1), torch.cat([o2, o2b], 1)
        else:
            out_1 = self.mod1(torch.cat([modal_1, modal_2], 1))
            out_2 = self.mod2_1(out_1)
        out_3 = self.mod3_1(out_2)
        out_4 = self.mod4_1(out_3)
        top = self.bn_out_1(self.mod5_1(out_4))
        seg, fusion, seg_cf = self.aspp(top)
        seg = self.fuse_seg(torch.cat([self.up_seg_2(seg), out_2], 1))
        seg_cert = None
        if self.cert_branch:  # :383-387
            seg_cert = torch.sigmoid(self.fuse_seg_cert(torch.cat([self.up_seg_2_cert(seg_cf), out_2], 1)))
        skip = F.adaptive_avg_pool2d(out_2, (fusion.size(2), fusion.size(3)))
        return seg, [seg, torch.cat([fusion, skip], 1), out_4, out_3, out_2, out_1], seg_cert


def cert_loss(pred_label_a, label_day, cert_a):
    """cm/train_trgb_segnet_conf.py:482-491 as written: the target 1 - p(true class) is [N, H, W] and NOT detached, the
    prediction is [N, 1, H, W]; their difference broadcasts to [N, N, H, W]."""
    one_hot = F.one_hot(label_day, pred_label_a.size(1)).permute(0, 3, 1, 2).to(pred_label_a.dtype)
    cert = 1.0 - torch.sum(one_hot * F.softmax(pred_label_a, dim=1), 1)
    return torch.mean((cert_a - cert) ** 2) * 10
