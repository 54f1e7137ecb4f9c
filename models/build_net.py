"""models/build_net.py of the reference: `models` table and `build_network(snapshot, backend) -> (net, epoch)`."""
from heatnet_amd.build_net import models  # noqa: F401
from heatnet_amd.build_net import build_network as _build


def build_network(snapshot, backend):
    return _build(snapshot, backend)
