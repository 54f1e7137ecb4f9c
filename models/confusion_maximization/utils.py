"""The hot-path helpers of models/confusion_maximization/utils.py (weights_init_normal, calculate_ious, loaders)."""
import numpy as np

from heatnet_amd import miou as _miou
from heatnet_amd.checkpoint import init_model_full as initModelFull  # noqa: F401
from heatnet_amd.checkpoint import init_model_partial as initModelPartial  # noqa: F401
from heatnet_amd.checkpoint import init_model_renamed as initModelRenamed  # noqa: F401
from heatnet_amd.conf_segnet import weights_init_normal  # noqa: F401


def calculate_ious(pred, target, n_classes=13):
    """utils.py:134-163 for tensors of class indices (integer counts formed on the host, as the reference does).
    For logits on the GPU use heatnet_amd.miou.confusion_matrix + ious_from_confusion instead."""
    pred = np.asarray(pred.detach().cpu() if hasattr(pred, "detach") else pred).reshape(-1).astype(np.int64)
    target = np.asarray(target.detach().cpu() if hasattr(target, "detach") else target).reshape(-1).astype(np.int64)
    conf = np.bincount(target * 14 + np.clip(pred, 0, 13), minlength=196).reshape(14, 14)
    return _miou.ious_from_confusion(conf, n_classes)
