"""cm/models/pspnet.py of the reference: HeatNet PSPNet, forward(modal_1, modal_2) -> (out, feats, None)."""
from heatnet_amd.pspnet import PSPModule, PSPNet, PSPUpsample  # noqa: F401
