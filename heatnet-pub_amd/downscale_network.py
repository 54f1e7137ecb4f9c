"""DownNet: brings the segmentation output down to a feature map's resolution so that it can be appended to a critic's
input (`--feedback_seg`).

models/confusion_maximization/models/downscale_network.py:5-35: `downsampling` x (conv3x3 without bias -> InstanceNorm2d ->
bilinear x0.5); module / state_dict names follow the reference (`net.<i>.conv1.weight`).

Note on the reference: `conv_segnet` builds these nets with the default `channels=12` while the PSPNet head emits 13
classes, so with `arch='pspnet'` a `feedback_seg` forward raises a channel-mismatch RuntimeError in the reference -- and
here (cm/models/conf_segnet.py:61-70,121-125; tests/test_gpu_network.py::test_feedback_seg_*).  The 12-class
configuration belongs to the ResNeXt 'custom' arch, which is outside this hot path; DownNet itself is complete.
"""
from collections import OrderedDict

import torch.nn as nn

from . import ops
from .layers import HnConv2d


class HnInstanceNorm2d(nn.InstanceNorm2d):
    """nn.InstanceNorm2d(C) as the reference uses it: no affine parameters, no running statistics, eps 1e-5."""

    def __init__(self, num_features, eps=1e-5):
        super().__init__(num_features, eps=eps, affine=False, track_running_stats=False)

    def forward(self, x):
        return ops.instance_norm(x, self.eps)


class Interpolate(nn.Module):
    def __init__(self, mode):
        super().__init__()
        if mode != "bilinear":
            raise NotImplementedError("Interpolate: the reference only uses mode='bilinear'")
        self.mode = mode

    def forward(self, x):
        h, w = x.size(2), x.size(3)
        if (h | w) & 1:
            # F.interpolate(scale_factor=0.5) samples at (o + 0.5) * 2 - 0.5 whatever the input size; the size-driven
            # kernel below agrees with that only for even sizes (all that the 320x704 / 640x1408 pipelines produce)
            raise NotImplementedError("Interpolate(x0.5): odd input sizes are not supported (got %dx%d)" % (h, w))
        return ops.bilinear(x, (h // 2, w // 2))


def convolveDownsample(channel_in, channel_out):
    return nn.Sequential(OrderedDict([("conv1", HnConv2d(channel_in, channel_out, 3, stride=1, padding=1, bias=False)),
                                      ("in1", HnInstanceNorm2d(channel_out)),
                                      ("downsample", Interpolate(mode="bilinear"))]))


class DownNet(nn.Module):
    def __init__(self, downsampling, channels=12):
        super().__init__()
        self.net = nn.Sequential(*[convolveDownsample(int(channels), int(channels)) for _ in range(downsampling)])

    def forward(self, seg):
        return self.net(seg)
