"""cm/models/build_net.py of the reference: build_network(snapshot, backend, in_channels, late_fusion) -> net."""
from heatnet_amd.build_net import build_network as _build


def build_network(snapshot, backend, in_channels, late_fusion):
    return _build(snapshot, backend, in_channels, late_fusion)
