"""models/extractors.py of the reference: dilated ResNets whose forward(x) returns (layer4, layer3) maps."""
from heatnet_amd import resnet as _r
from heatnet_amd.resnet import BasicBlock, Bottleneck, model_urls  # noqa: F401


def resnet18(pretrained=True):
    return _r.resnet18(pretrained, dual=False)


def resnet34(pretrained=True):
    return _r.resnet34(pretrained, dual=False)


def resnet50(pretrained=True):
    return _r.resnet50(pretrained, dual=False)


def resnet101(pretrained=True):
    return _r.resnet101(pretrained, dual=False)


def resnet152(pretrained=True):
    return _r.resnet152(pretrained, dual=False)
