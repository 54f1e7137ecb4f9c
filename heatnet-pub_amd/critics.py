"""Fully-convolutional day/night critics on the HIP kernels.

models/confusion_maximization/discriminator_model.py:35-64: five 4x4 stride-2 convolutions (LeakyReLU 0.2 fused in
the conv epilogue of the first four) and a bilinear x32 upsample of the one-channel map.
"""
import torch.nn as nn

from . import ops
from .layers import HnConv2d


class FCDiscriminator(nn.Module):
    def __init__(self, num_classes, ndf=64):
        super().__init__()
        self.conv1 = HnConv2d(num_classes, ndf, kernel_size=4, stride=2, padding=1)
        self.conv2 = HnConv2d(ndf, ndf * 2, kernel_size=4, stride=2, padding=1)
        self.conv3 = HnConv2d(ndf * 2, ndf * 4, kernel_size=4, stride=2, padding=1)
        self.conv4 = HnConv2d(ndf * 4, ndf * 8, kernel_size=4, stride=2, padding=1)
        self.classifier = HnConv2d(ndf * 8, 1, kernel_size=4, stride=2, padding=1)
        self.negative_slope = 0.2
        self.scale_factor = 32

    def forward(self, x):
        for conv in (self.conv1, self.conv2, self.conv3, self.conv4):
            x = conv(x, act=ops.ACT_LEAKY, slope=self.negative_slope)
        x = self.classifier(x)
        return ops.bilinear(x, (x.size(2) * self.scale_factor, x.size(3) * self.scale_factor))
