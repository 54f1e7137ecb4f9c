"""mIoU of the reference's two evaluators from one on-device argmax + confusion-matrix histogram.

    models/confusion_maximization/utils.py:134-163  calculate_ious  (13 classes, class 12 and label 13 excluded)
    scripts/iou_eval.py:103-182                      IoU(num_classes, ignore_index).value()
Integer counts come from `hn_argmax_confusion`; the per-class ratios are formed on the host in float64 exactly as
the reference does.
"""
import numpy as np

from . import ops


def confusion_matrix(logits, labels, num_classes):
    """int64 [K,K] counts, rows = label, columns = prediction; labels outside [0,K) are not counted."""
    _, conf = ops.argmax_confusion(logits, labels, num_classes)
    return conf.cpu().numpy()


def ious_from_confusion(conf, n_classes=13, background_class_idx=12, ignore_class_idx=13):
    """`calculate_ious` from a (>=14)x(>=14) confusion matrix (label 13 = ignore must have its own row)."""
    conf = np.asarray(conf, dtype=np.int64)
    keep_rows = [r for r in range(conf.shape[0]) if r != ignore_class_idx]
    ious = []
    for cls in range(n_classes):
        if cls in (background_class_idx, ignore_class_idx):
            continue
        inter = int(conf[cls, cls])
        pred_cnt = int(conf[keep_rows, cls].sum())
        tgt_cnt = int(conf[cls, :].sum())
        union = pred_cnt + tgt_cnt - inter
        ious.append(float("nan") if union == 0 else inter / max(union, 1))
    return np.array(ious)


def iou_eval_value(conf, ignore_index=()):
    """`IoU.value()` of scripts/iou_eval.py: per-class IoU and nanmean."""
    conf = np.array(conf, dtype=np.int64)
    for idx in ignore_index:
        conf[:, idx] = 0
        conf[idx, :] = 0
    tp = np.diag(conf)
    fp = conf.sum(0) - tp
    fn = conf.sum(1) - tp
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (tp + fp + fn)
    return iou, np.nanmean(iou)
