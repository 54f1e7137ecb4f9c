"""models/confusion_maximization/discriminator_model.py of the reference: the fully-convolutional critic."""
from heatnet_amd.critics import FCDiscriminator  # noqa: F401
