"""The reference's ResNeXt-50 + ASPP "custom" segmentation network on the HIP kernels.

Operator surface and `state_dict` keys of models/confusion_maximization/models/trgb_segnet.py:242-402 (`ResNeXt`), the
default `arch` of `conv_segnet` (conf_segnet.py:23,38-43): pre-activation bottlenecks with 64-group 3x3 convolutions
(`IdentityResidualBlock`, :47-129), ASPP with dilation 12 / 24 / 36 and a global-average branch (:133-183), a decoder of
transposed convolutions (:201-209, 338).  An optional branch of the hot path: built from the same
operators as the PSPNet path -- conv -> BatchNorm -> ReLU units run through `layers.conv_bn_act` (the fused f16x2 node where it
covers the layer) -- plus three of its own: the grouped 3x3 convolution (csrc/grouped_conv.hip), the transposed convolution
(the data-gradient kernels run forwards) and the residual add.
"""
from collections import OrderedDict

import torch.nn as nn

from . import ops
from .layers import HnBatchNorm2d, HnConv2d, HnMaxPool3x3s2, _NoDataParallel, conv_bn_act


class ABN(nn.Sequential):
    """BatchNorm2d + ReLU (:24-44): keys `bn.*`; applied as one fused kernel."""

    def __init__(self, num_features):
        super().__init__(OrderedDict([("bn", HnBatchNorm2d(num_features)), ("act", nn.ReLU(inplace=True))]))

    def forward(self, x):
        return self.bn(ops.as_f32(x), act=ops.ACT_RELU)


class HnGroupedConv3x3(_NoDataParallel, nn.Conv2d):
    """nn.Conv2d(C, C, 3, padding=d, dilation=d, groups=G, bias=False) with 2 .. 16 channels per group."""

    def __init__(self, channels, groups, dilation):
        super().__init__(channels, channels, 3, stride=1, padding=dilation, dilation=dilation, groups=groups, bias=False)
        import torch
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        return ops.grouped_conv3x3(ops.as_f32(x), self.weight, self.groups, self.dilation[0])


class HnConvTranspose2d(_NoDataParallel, nn.ConvTranspose2d):
    """nn.ConvTranspose2d(Cin, Cout, K, stride, padding, bias=False)."""

    def __init__(self, cin, cout, k, stride, padding):
        super().__init__(cin, cout, k, stride, padding, bias=False)
        import torch
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def forward(self, x):
        return ops.conv_transpose2d(ops.as_f32(x), self.weight, self.stride[0], self.padding[0])


class IdentityResidualBlock(nn.Module):
    """Pre-activation bottleneck (:47-129): bn1(x) -> conv1 1x1 (stride) -> bn2 -> conv2 3x3 grouped (dilation) -> bn3 -> conv3
    1x1, plus x (or proj_conv(bn1(x)) where the shape changes)."""

    def __init__(self, in_channels, channels, stride=1, dilation=1, groups=64):
        super().__init__()
        c1, c2, c3 = channels
        self.bn1 = ABN(in_channels)
        self.convs = nn.Sequential(OrderedDict([
            ("conv1", HnConv2d(in_channels, c1, 1, stride=stride, padding=0, bias=False)),
            ("bn2", ABN(c1)),
            ("conv2", HnGroupedConv3x3(c2, groups, dilation)),
            ("bn3", ABN(c2)),
            ("conv3", HnConv2d(c2, c3, 1, stride=1, padding=0, bias=False))]))
        if stride != 1 or in_channels != c3:
            self.proj_conv = HnConv2d(in_channels, c3, 1, stride=stride, padding=0, bias=False)

    def forward(self, x):
        a = self.bn1(x)
        shortcut = self.proj_conv(a) if hasattr(self, "proj_conv") else x
        y = conv_bn_act(a, self.convs.conv1, self.convs.bn2.bn, ops.ACT_RELU)
        y = self.convs.bn3(self.convs.conv2(y))
        return ops.add(self.convs.conv3(y), shortcut)


class UnitABN(nn.Sequential):
    """`createConv` of the ASPP head (:154-166): conv (no bias) -> ABN; keys conv1.weight, bn1.bn.*"""

    def __init__(self, cin, cout, k, dil):
        pad = dil if dil != 1 else (1 if k == 3 else 0)
        super().__init__(OrderedDict([("conv1", HnConv2d(cin, cout, k, stride=1, padding=pad, dilation=dil, bias=False)),
                                      ("bn1", ABN(cout))]))

    def forward(self, x):
        return conv_bn_act(x, self.conv1, self.bn1.bn, ops.ACT_RELU)


class ASPP(nn.Module):
    def __init__(self, in_channels, out_channels, rates=(1, 12, 24, 36), kernels=(1, 3, 3, 3)):
        super().__init__()
        for i in range(4):
            setattr(self, "conv%d" % (i + 1), UnitABN(in_channels, 256, kernels[i], rates[i]))
        self.conv5 = UnitABN(in_channels, 256, 1, 1)
        self.fuse_conv = UnitABN(256 * 5, 256, 1, 1)
        self.final_conv = UnitABN(256, out_channels, 1, 1)
        self.final_conv2 = UnitABN(256, 1, 1, 1)
        self.cert_head = False

    def forward(self, x):
        xs = [getattr(self, "conv%d" % i)(x) for i in range(1, 5)]
        g = self.conv5(ops.adaptive_avg_pool(ops.as_f32(x), 1))
        xs.append(ops.bilinear(g, (x.size(2), x.size(3))))  # (a 1 x 1 map: every resampling convention broadcasts it)
        fusion = self.fuse_conv(ops.concat(xs))
        out_cert = None
        if self.cert_head:  # read by the certainty branch of the network (:383-387)
            out_cert = self.final_conv2(fusion)
        elif self.training:
            # the reference always evaluates the certainty head (:180) although only `cert_branch` reads it: its BatchNorm
            # running statistics move with every training forward -- kept so that state_dicts stay interchangeable
            import torch
            with torch.no_grad():
                self.final_conv2(fusion.detach())
        return self.final_conv(fusion), fusion, out_cert


def init_block(input_channels):
    return nn.Sequential(OrderedDict([
        ("conv1_2", HnConv2d(input_channels, 64, 3, stride=2, padding=1, bias=False)), ("bn1_2", ABN(64)),
        ("conv2", HnConv2d(64, 64, 3, stride=1, padding=1, bias=False)), ("bn2", ABN(64)),
        ("conv3", HnConv2d(64, 64, 3, stride=1, padding=1, bias=False)), ("pool", HnMaxPool3x3s2())]))


def run_init_block(m, x):
    y = conv_bn_act(x, m.conv1_2, m.bn1_2.bn, ops.ACT_RELU)
    y = conv_bn_act(y, m.conv2, m.bn2.bn, ops.ACT_RELU)
    return m.pool(m.conv3(y))


def fuse_module(channel_in, channel_out):
    return nn.Sequential(OrderedDict([
        ("conv1", HnConv2d(channel_in, channel_in, 3, stride=1, padding=1, bias=False)), ("bn1", ABN(channel_in)),
        ("conv2", HnConv2d(channel_in, channel_in, 3, stride=1, padding=1, bias=False)), ("bn2", ABN(channel_in)),
        ("conv3", HnConv2d(channel_in, channel_out, 3, stride=1, padding=1, bias=False)),
        ("up", HnConvTranspose2d(channel_out, channel_out, 8, 4, 2))]))


def run_fuse_module(m, x):
    y = conv_bn_act(x, m.conv1, m.bn1.bn, ops.ACT_RELU)
    y = conv_bn_act(y, m.conv2, m.bn2.bn, ops.ACT_RELU)
    return m.up(m.conv3(y))


class ResNeXt(nn.Module):
    """`ResNeXt(structure, groups=64, classes=13, dilation=[1,1,2,4], base_channels=(128,128,256), input_channels=4,
    cert_branch=False, late_fusion=False)` -> forward(modal_1, modal_2) = (seg, [seg, cat(fusion, pooled out_2), out_4, out_3,
    out_2, out_1], seg_cert or None)."""

    def __init__(self, structure, groups=64, norm_act=None, classes=13, dilation=(1, 1, 2, 4), base_channels=(128, 128, 256),
                 input_channels=4, cert_branch=False, late_fusion=False):
        super().__init__()
        if len(structure) != 4 or len(dilation) != 4:
            raise ValueError("Expected a structure and a dilation list with four values")
        self.structure, self.late_fusion = structure, late_fusion
        if late_fusion:
            self.mod1, self.mod1_2 = init_block(3), init_block(1)
        else:
            self.mod1 = init_block(input_channels)
        in_channels, channels = 64, list(base_channels)
        for mod_id, num in enumerate(structure):
            start = in_channels
            for rep in range(2 if (late_fusion and mod_id == 0) else 1):
                in_channels = start * 2 if (mod_id == 1 and late_fusion) else start
                blocks = []
                for block_id in range(num):
                    s, d = self._stride_dilation(mod_id, block_id, dilation)
                    blocks.append(("block%d" % (block_id + 1), IdentityResidualBlock(in_channels, channels, s, d, groups)))
                    in_channels = channels[-1]
                self.add_module("mod%d_%d" % (mod_id + 2, rep + 1), nn.Sequential(OrderedDict(blocks)))
            channels = [c * 2 for c in channels]
        self.bn_out_1, self.bn_out_2 = ABN(in_channels), ABN(in_channels)
        self.aspp = ASPP(in_channels, classes)
        self.up_seg_2 = HnConvTranspose2d(classes, classes, 4, 2, 1)
        self.up_borders_3 = HnConvTranspose2d(5, 5, 4, 2, 1)  # unused heads of the reference: kept for the state_dict keys
        self.up_inst_2 = HnConvTranspose2d(5, 5, 4, 2, 1)
        self.fuse_seg = fuse_module((512 if late_fusion else 256) + classes, classes)
        self.cert_branch = cert_branch
        if cert_branch:  # (:344-346; sized for the early-fusion skip tensor of 256 channels, as in the reference)
            self.up_seg_2_cert = HnConvTranspose2d(1, 1, 4, 2, 1)
            self.fuse_seg_cert = fuse_module(256 + 1, 1)
            self.aspp.cert_head = True
        self.forward_decoder = True

    def setForwardDecoder(self, state):
        self.forward_decoder = state

    @staticmethod
    def _stride_dilation(mod_id, block_id, dilation):
        if dilation[mod_id] == 1:
            return (2 if (mod_id > 0 and block_id == 0) else 1), 1
        return 1, dilation[mod_id]

    def forward(self, modal_1, modal_2=None):
        if self.late_fusion and modal_2 is not None:
            if ops.DUAL_STREAMS and modal_1.is_cuda and ops._conv_prof is None:
                # the thermal chain (mod1_2, mod2_2) on a second stream, as in resnet.ResNet.forward (ops.DUAL_STREAMS)
                import torch
                main, s2 = ops.dual_fork(modal_1.device)
                with torch.cuda.stream(s2):
                    o1b = run_init_block(self.mod1_2, modal_2)
                    o2b = self.mod2_2(o1b)
                o1 = run_init_block(self.mod1, modal_1)
                o2 = self.mod2_1(o1)
                ops.dual_join(main, s2)
                ops.hand_over(o1b, main)
                ops.hand_over(o2b, main)
            else:
                o1, o1b = run_init_block(self.mod1, modal_1), run_init_block(self.mod1_2, modal_2)
                o2, o2b = self.mod2_1(o1), self.mod2_2(o1b)
            out_1, out_2 = ops.concat([o1, o1b]), ops.concat([o2, o2b])
        else:
            out_1 = run_init_block(self.mod1, ops.concat([modal_1, modal_2]))
            out_2 = self.mod2_1(out_1)
        out_3 = self.mod3_1(out_2)
        out_4 = self.mod4_1(out_3)
        top = self.bn_out_1(self.mod5_1(out_4))
        seg, fusion, seg_cf = self.aspp(top)
        seg_cert = None
        if self.forward_decoder:
            seg = run_fuse_module(self.fuse_seg, ops.concat([self.up_seg_2(seg), out_2]))
            if self.cert_branch:  # (:383-387)
                if out_2.size(1) != 256:
                    raise RuntimeError("cert_branch: fuse_seg_cert takes 256 + 1 channels, the late-fusion skip tensor has %d "
                                       "(the reference fails the same way)" % out_2.size(1))
                seg_cert = run_fuse_module(self.fuse_seg_cert, ops.concat([self.up_seg_2_cert(seg_cf), out_2]))
                seg_cert = ops.sigmoid(seg_cert)
        else:
            seg = None
        skip = ops.adaptive_avg_pool(out_2, (fusion.size(2), fusion.size(3)))
        return seg, [seg, ops.concat([fusion, skip]), out_4, out_3, out_2, out_1], seg_cert


def net_resnext50(**kw):
    return ResNeXt(structure=[3, 4, 6, 3], **kw)


def net_resnext101(**kw):  # (`_NETS`, :419-429)
    return ResNeXt(structure=[3, 4, 23, 3], **kw)


def net_resnext152(**kw):
    return ResNeXt(structure=[3, 8, 36, 3], **kw)
