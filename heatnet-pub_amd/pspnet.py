"""PSPNet head and decoder of HeatNet on the HIP kernels.

Operator surface and `state_dict` keys follow
    models/confusion_maximization/models/pspnet.py:8-76   (HeatNet: forward(modal_1, modal_2) -> (out, feats, None))
    models/pspnet.py:8-75                                 (RGB-only: forward(x) -> logits, 18 classes)
"""
import torch.nn as nn

from . import ops
from . import resnet as extractors
from .layers import HnAdaptiveAvgPool2d, HnBatchNorm2d, HnConv2d, HnDropout2d, HnPReLU, conv_bn_act


class PSPModule(nn.Module):
    """Pyramid pooling: 4 x (adaptive avg pool -> 1x1 conv -> bilinear up), concat with the input, 1x1 + ReLU."""

    def __init__(self, features, out_features=1024, sizes=(1, 2, 3, 6)):
        super().__init__()
        self.stages = nn.ModuleList([self._make_stage(features, size) for size in sizes])
        self.bottleneck = HnConv2d(features * (len(sizes) + 1), out_features, kernel_size=1)
        self.relu = nn.ReLU()

    @staticmethod
    def _make_stage(features, size):
        return nn.Sequential(HnAdaptiveAvgPool2d((size, size)), HnConv2d(features, features, kernel_size=1, bias=False))

    def forward(self, feats):
        """relu(bottleneck(cat([up(stage_i(feats))..., feats]))) without building the 5*C-channel concatenation.

        A 1x1 convolution commutes with bilinear resampling, so the bottleneck's weight columns that belong to
        pyramid level i are applied to the pooled s x s map BEFORE it is upsampled; one fused kernel then adds the
        four upsampled C_out-channel maps to the full-resolution term and applies the ReLU.  Same parameters, same
        function, 1/5 of the bottleneck's multiply-adds at full resolution."""
        w = self.bottleneck.weight
        f = feats.size(1)
        k = len(self.stages)
        if w.is_cuda and w.size(1) == (k + 1) * f:
            cols = ops.split_columns(w, [f] * (k + 1))  # one autograd node: the slices' gradients land in ONE full-size dW
        else:
            cols = [w[:, i * f:(i + 1) * f] for i in range(k + 1)]
        if feats.requires_grad:
            # the pooled branches read the main conv's pass-through alias: their (single, fused) gradient enters the
            # main conv's dgrad kernel as its addend instead of being summed by autograd
            main, feats = ops.conv2d(feats, cols[k], self.bottleneck.bias, passthrough=True)
        else:
            main = ops.conv2d(feats, cols[k], self.bottleneck.bias)
        pooled = ops.pyramid_pool(feats, [stage[0].output_size for stage in self.stages])
        levels = []
        for i, stage in enumerate(self.stages):
            q = stage[1](pooled[i])                  # 1x1 conv (C -> C) at s x s
            levels.append(ops.conv2d(q, cols[i]))    # this level's slice of the bottleneck, still s x s
        return ops.pyramid_merge(main, levels)


class PSPUpsample(nn.Module):
    """bilinear x2 -> conv3x3(+bias) -> BatchNorm -> PReLU (one fused BN/PReLU kernel)."""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.conv = nn.Sequential(HnConv2d(in_channels, out_channels, 3, padding=1), HnBatchNorm2d(out_channels),
                                  HnPReLU())

    def forward(self, x, drop=None):
        """`drop`: the Dropout2d layer that follows this block in PSPNet; its channel mask is applied in the BN/PReLU
        kernel's output pass instead of by a pass of its own."""
        conv, bn = self.conv[0], self.conv[1]
        n, c, h, w = x.shape
        s_, p_, d_ = conv.geometry()
        if ops.fused_conv_bn_ok(x, conv.weight, s_, p_, d_, bn.training or bn.running_mean is None, shape=(n, c, 2 * h, 2 * w)):
            p = ops.bilinear_up2_planes(x)  # read by the convolution only: written as its f16x2 operand
        else:
            p = ops.bilinear(x, (2 * h, 2 * w))
        mask = drop.mask_for(n, conv.out_channels, x.device) if drop is not None else None
        mask_max = 1.0 / (1.0 - drop.p) if mask is not None else 1.0
        return conv_bn_act(p, conv, bn, ops.ACT_PRELU, prelu_weight=self.conv[2].weight, drop_mask=mask, mask_max=mask_max)


class PSPNet(nn.Module):
    """`rgb_only=False`: HeatNet signature; `rgb_only=True`: the top-level models/pspnet.py signature."""

    def __init__(self, n_classes=13, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet34",
                 pretrained=True, late_fusion=False, in_channels=3, rgb_only=False):
        super().__init__()
        self.rgb_only = rgb_only
        ctor = getattr(extractors, backend)
        if rgb_only:
            self.feats = ctor(pretrained, dual=False)
        elif backend == "resnet50":
            self.feats = ctor(pretrained, late_fusion=late_fusion, in_channels=in_channels)
        else:
            self.feats = ctor(pretrained)
        self.psp = PSPModule(psp_size, 1024, sizes)
        self.drop_1 = HnDropout2d(p=0.3)
        self.up_1 = PSPUpsample(1024, 256)
        self.up_2 = PSPUpsample(256, 64)
        self.up_3 = PSPUpsample(64, 64)
        self.drop_2 = HnDropout2d(p=0.15)
        self.final = nn.Sequential(HnConv2d(64, n_classes, kernel_size=1))

    def forward(self, modal_1, modal_2=None):
        if self.rgb_only:
            top, _ = self.feats(modal_1)
            f = None
        else:
            f = self.feats(modal_1, modal_2)
            top = f[0]
        p = self.drop_1(self.psp(top))
        p = self.up_1(p, self.drop_2)  # drop_2 folded into each decoder block's BN/PReLU pass
        p = self.up_2(p, self.drop_2)
        p = self.up_3(p, self.drop_2)
        out = self.final(p)
        if self.rgb_only:
            return out
        return out, [out, f[0], f[1], f[2], f[3], f[4]], None
