"""models/pspnet.py of the reference (RGB-only PSPNet: forward(x) -> logits, 18 classes by default)."""
from heatnet_amd.pspnet import PSPModule, PSPUpsample  # noqa: F401
from heatnet_amd.pspnet import PSPNet as _PSPNet


class PSPNet(_PSPNet):
    def __init__(self, n_classes=18, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet34",
                 pretrained=True):
        super().__init__(n_classes, sizes, psp_size, deep_features_size, backend, pretrained, rgb_only=True)
