"""cm/models/downscale_network.py of the reference: DownNet (segmentation output -> critic-input resolution)."""
from heatnet_amd.downscale_network import DownNet, Interpolate, convolveDownsample  # noqa: F401
