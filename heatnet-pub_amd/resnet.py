"""Dilated ResNet encoders (output stride 8) of HeatNet on the HIP kernels.

Mirrors the operator surface and `state_dict` keys of the reference encoders:
    models/confusion_maximization/models/extractors.py:29-198  (dual-stream: `forward(modal_1, modal_2)` -> 5 maps)
    models/extractors.py:29-154                                (RGB-only:   `forward(x)` -> (x, x_3))
Every residual block runs as conv(+batch-stat epilogue) -> fused BN/ReLU(/add) kernels.
"""
import math

import torch.nn as nn
from torch.utils import model_zoo

from . import ops
from .layers import HnBatchNorm2d, HnConv2d, HnMaxPool3x3s2, conv_bn_act

model_urls = {
    "resnet18": "https://download.pytorch.org/models/resnet18-5c106cde.pth",
    "resnet34": "https://download.pytorch.org/models/resnet34-333f7ec4.pth",
    "resnet50": "https://download.pytorch.org/models/resnet50-19c8e357.pth",
    "resnet101": "https://download.pytorch.org/models/resnet101-5d3b4d8f.pth",
    "resnet152": "https://download.pytorch.org/models/resnet152-b121ed2d.pth",
}


def _shortcut(block, x):
    if block.downsample is None:
        return x
    return conv_bn_act(x, block.downsample[0], block.downsample[1], ops.ACT_NONE)


def _first_conv(block, x):
    """conv1 + bn1 + ReLU of a residual block.  When a gradient will flow, `x` is re-bound to conv1's pass-through
    alias, so the skip connection's gradient is added inside conv1's dgrad kernel rather than by a separate pass.
    The result is only ever read by conv2: a plane tensor where the fused f16x2 node covers the layer."""
    if x.requires_grad:
        return conv_bn_act(x, block.conv1, block.bn1, ops.ACT_RELU, passthrough=True, out="planes")
    return conv_bn_act(x, block.conv1, block.bn1, ops.ACT_RELU, out="planes"), x


def run_layer(layer, x):
    """nn.Sequential of residual blocks: the tensors BETWEEN the blocks are read by the next block only (its conv1, its
    residual add, its downsample conv) and travel as plane tensors; the layer's output is an fp32 tensor with a plane twin."""
    blocks = list(layer)
    for i, blk in enumerate(blocks):
        x = blk(x, out="planes" if i + 1 < len(blocks) else "dual")
    return x


class BasicBlock(nn.Module):
    """Two 3x3 convs (ResNet-18/34), models/extractors.py:34-63."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = HnConv2d(inplanes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn1 = HnBatchNorm2d(planes)
        self.conv2 = HnConv2d(planes, planes, 3, stride=1, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = HnBatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, out="f32"):
        y, x = _first_conv(self, x)
        return conv_bn_act(y, self.conv2, self.bn2, ops.ACT_RELU, residual=_shortcut(self, x), out=out)


class Bottleneck(nn.Module):
    """1x1 -> 3x3 (stride, dilation) -> 1x1, models/confusion_maximization/models/extractors.py:66-102."""
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, dilation=1):
        super().__init__()
        self.conv1 = HnConv2d(inplanes, planes, 1, bias=False)
        self.bn1 = HnBatchNorm2d(planes)
        self.conv2 = HnConv2d(planes, planes, 3, stride=stride, dilation=dilation, padding=dilation, bias=False)
        self.bn2 = HnBatchNorm2d(planes)
        self.conv3 = HnConv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = HnBatchNorm2d(planes * 4)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, out="f32"):
        y, x = _first_conv(self, x)
        y = conv_bn_act(y, self.conv2, self.bn2, ops.ACT_RELU, out="planes")
        return conv_bn_act(y, self.conv3, self.bn3, ops.ACT_RELU, residual=_shortcut(self, x), out=out)


class ResNet(nn.Module):
    """Dual-stream capable dilated ResNet.

    `dual=True` gives the HeatNet interface (conf-max extractors.py:172-198): `forward(modal_1, modal_2)`
    returning [x_5, x_4, x_3, x_2, x_1] with the IR stream (conv1_2/bn1_2/layer1_2/layer2_2) concatenated in
    front of layer3 when `late_fusion`.  `dual=False` gives the RGB-only interface of models/extractors.py:143-154:
    `forward(x)` returning (x, x_3).
    """

    def __init__(self, block, layers=(3, 4, 23, 3), late_fusion=False, in_channels=3, dual=True):
        super().__init__()
        self.inplanes = 64
        self.late_fusion = late_fusion
        self.dual = dual
        self.conv1 = HnConv2d(3 if late_fusion else in_channels, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = HnBatchNorm2d(64)
        if late_fusion:
            self.conv1_2 = HnConv2d(1, 64, 7, stride=2, padding=3, bias=False)
            self.bn1_2 = HnBatchNorm2d(64)
        self.maxpool = HnMaxPool3x3s2()

        keep = self.inplanes
        self.layer1 = self._make_layer(block, 64, layers[0])
        if late_fusion:
            self.inplanes = keep
            self.layer1_2 = self._make_layer(block, 64, layers[0])
        keep = self.inplanes
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        if late_fusion:
            self.inplanes = keep
            self.layer2_2 = self._make_layer(block, 128, layers[1], stride=2)
            self.inplanes *= 2
        self.layer3 = self._make_layer(block, 256, layers[2], stride=1, dilation=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=1, dilation=4)

        for m in self.modules():  # conf-max extractors.py:148-154
            if isinstance(m, nn.Conv2d):
                fan = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / fan))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1, dilation=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                HnConv2d(self.inplanes, planes * block.expansion, 1, stride=stride, bias=False),
                HnBatchNorm2d(planes * block.expansion))
        stages = [block(self.inplanes, planes, stride, downsample)]  # first block: dilation 1
        self.inplanes = planes * block.expansion
        stages += [block(self.inplanes, planes, dilation=dilation) for _ in range(1, blocks)]
        return nn.Sequential(*stages)

    def _stem(self, conv, bn, x):
        return self.maxpool(conv_bn_act(x, conv, bn, ops.ACT_RELU))

    def forward(self, modal_1, modal_2=None):
        if not self.late_fusion and modal_2 is not None:
            modal_1 = ops.concat([modal_1, modal_2])  # early fusion: one 4-channel stem
        # (not while bench.py's instrumented pass brackets every conv launch with events: overlapped durations would not add up)
        dual = ops.DUAL_STREAMS and self.late_fusion and modal_2 is not None and modal_1.is_cuda and ops._conv_prof is None
        if dual:
            # the thermal encoder (stem, layer1_2, layer2_2) is independent of the RGB one up to the fusion point: it runs on a
            # second stream (ops.DUAL_STREAMS); autograd replays each node's backward on its forward stream
            import torch
            main, s2 = ops.dual_fork(modal_1.device)
            with torch.cuda.stream(s2):
                t_1 = self._stem(self.conv1_2, self.bn1_2, modal_2)
                t_2 = run_layer(self.layer1_2, t_1)
                t_3 = run_layer(self.layer2_2, t_2)
        x_1 = self._stem(self.conv1, self.bn1, modal_1)
        x_2 = run_layer(self.layer1, x_1)
        x_3 = run_layer(self.layer2, x_2)
        if dual:
            ops.dual_join(main, s2)
            for t in (t_1, t_2, t_3):
                ops.hand_over(t, main)
            x_3 = ops.concat([x_3, t_3])
        elif self.late_fusion and modal_2 is not None:
            t_1 = self._stem(self.conv1_2, self.bn1_2, modal_2)
            t_2 = run_layer(self.layer1_2, t_1)
            t_3 = run_layer(self.layer2_2, t_2)
            x_3 = ops.concat([x_3, t_3])
        x_4 = run_layer(self.layer3, x_3)
        x_5 = run_layer(self.layer4, x_4)
        if not self.dual:
            return x_5, x_4
        if self.late_fusion:
            return [x_5, x_4, x_3, ops.concat([x_2, t_2]), ops.concat([x_1, t_1])]
        return [x_5, x_4, x_3, x_2, x_1]


def _load_sequential(target, source_state):
    own = target.state_dict()
    target.load_state_dict({k: v for k, v in source_state.items() if k in own})


def _build(name, block, layers, pretrained, **kw):
    model = ResNet(block, layers, **kw)
    if pretrained:  # same URL table as the reference; needs network access, exactly like the reference
        _load_sequential(model, model_zoo.load_url(model_urls[name]))
    return model


def resnet18(pretrained=True, **kw):
    return _build("resnet18", BasicBlock, [2, 2, 2, 2], pretrained, **kw)


def resnet34(pretrained=True, **kw):
    return _build("resnet34", BasicBlock, [3, 4, 6, 3], pretrained, **kw)


def resnet50(pretrained=True, late_fusion=False, in_channels=3, **kw):
    return _build("resnet50", Bottleneck, [3, 4, 6, 3], pretrained, late_fusion=late_fusion, in_channels=in_channels, **kw)


def resnet101(pretrained=True, **kw):
    return _build("resnet101", Bottleneck, [3, 4, 23, 3], pretrained, **kw)


def resnet152(pretrained=True, **kw):
    return _build("resnet152", Bottleneck, [3, 8, 36, 3], pretrained, **kw)
