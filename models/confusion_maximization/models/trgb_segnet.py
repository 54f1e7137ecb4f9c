"""cm/models/trgb_segnet.py of the reference: the ResNeXt + ASPP "custom" segmentation network (`arch='custom'`)."""
from heatnet_amd.trgb_segnet import (ABN, ASPP, IdentityResidualBlock, ResNeXt, net_resnext50, net_resnext101,  # noqa: F401
                                     net_resnext152)
