"""Model-construction boundary: both `build_network` signatures of the reference.

    models/confusion_maximization/models/build_net.py:18-28   build_network(snapshot, backend, in_channels, late_fusion) -> net
    models/build_net.py:7-28                                  build_network(snapshot, backend) -> (net, epoch)
Snapshots are files named `<name>_<epoch>` holding a `state_dict` (reference build_net.py:24-26).
"""
import os

import torch

from .layers import ensure_channels_last_
from .pspnet import PSPNet

# models/build_net.py:7-15 -- the squeezenet / densenet rows need torchvision weights and are outside the hot path
models = {
    "resnet18": lambda: PSPNet(n_classes=18, sizes=(1, 2, 3, 6), psp_size=512, deep_features_size=256, backend="resnet18", rgb_only=True),
    "resnet34": lambda: PSPNet(n_classes=18, sizes=(1, 2, 3, 6), psp_size=512, deep_features_size=256, backend="resnet34", rgb_only=True),
    "resnet50": lambda: PSPNet(n_classes=18, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet50", rgb_only=True),
    "resnet101": lambda: PSPNet(n_classes=18, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet101", rgb_only=True),
    "resnet152": lambda: PSPNet(n_classes=18, sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet152", rgb_only=True),
}


def _load_snapshot(net, snapshot):
    _, epoch = os.path.basename(snapshot).split("_")
    net.load_state_dict(torch.load(snapshot))
    return int(epoch)


def _to_device(net):
    # the reference ends with `net.cuda()`; on a host without a GPU the module is returned on the CPU, where it
    # can be inspected / (de)serialised but every forward raises (the operators have no CPU path)
    if torch.cuda.is_available():
        net = net.cuda()
    return ensure_channels_last_(net)


def build_network(snapshot, backend, in_channels=None, late_fusion=None):
    """Four arguments: the HeatNet factory (returns `net`).  Two arguments: the RGB-only factory (`(net, epoch)`)."""
    backend = backend.lower()
    if in_channels is None and late_fusion is None:
        net, epoch = models[backend](), 0
        if snapshot is not None:
            epoch = _load_snapshot(net, snapshot)
        return _to_device(net), epoch
    # the reference ignores `backend` here and always builds the ResNet-50 PSPNet without pretrained weights
    net = PSPNet(sizes=(1, 2, 3, 6), psp_size=2048, deep_features_size=1024, backend="resnet50",
                 in_channels=in_channels, pretrained=False, late_fusion=bool(late_fusion))
    if snapshot is not None:
        _load_snapshot(net, snapshot)
    return _to_device(net)
