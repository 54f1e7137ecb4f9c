"""cm/models/extractors.py of the reference: dual-stream dilated ResNets."""
from heatnet_amd.resnet import BasicBlock, Bottleneck, ResNet, model_urls  # noqa: F401
from heatnet_amd.resnet import resnet18, resnet34, resnet50, resnet101, resnet152  # noqa: F401
