"""cm/models/conf_segnet.py of the reference: the segnet + critics wrapper driven by the HeatNet trainer."""
from heatnet_amd.conf_segnet import conv_segnet, create_critic  # noqa: F401
