"""Validation with the reference's conventions, mIoU accumulated on the device.

models/confusion_maximization/validation_bdd_mf.py:259-379 (`validate_model`): every sample is DUPLICATED to a batch
of two (:297-299), pushed through the model under `no_grad` WITHOUT `model.eval()` (:263 is commented out, so
BatchNorm normalises with the statistics of that two-sample batch and updates its running buffers, and Dropout2d is
active), row 0 is kept and arg-maxed (:301-335), and `calculate_ious` runs over all predictions (:361).
`reference_mode=False` is the conventional alternative (eval mode, any batch size).  The argmax and the confusion
histogram run in `hn_argmax_confusion`; only the 14x14 int64 matrix leaves the GPU.
"""
import numpy as np
import torch

from . import miou, ops


def _forward(model, rgb, ir):
    out = model(rgb, ir) if ir is not None else model(rgb)
    return out[0] if isinstance(out, tuple) else out


def validate_model(model, samples, reference_mode=True, num_classes=14):
    """`samples` yields (rgb [B,3,H,W], ir [B,1,H,W] or None, label [B,H,W] int64) on the model's device.

    Returns {"confusion": int64 [K,K] (rows = label), "ious": calculate_ious array, "miou": nanmean of it,
             "iou_eval": (per-class IoU, mIoU) of scripts/iou_eval.IoU(14, ignore_index=[12, 13])}.
    """
    was_training = model.training
    conf = None
    try:
        if reference_mode:
            model.train()  # the reference never leaves train mode during validation
        else:
            model.eval()
        with torch.no_grad(), ops.frozen_weights():  # evaluation: weight operands prepared once for the whole loop
            for rgb, ir, label in samples:
                if reference_mode:
                    for b in range(rgb.shape[0]):
                        r2 = rgb[b:b + 1].expand(2, -1, -1, -1).contiguous()
                        i2 = ir[b:b + 1].expand(2, -1, -1, -1).contiguous() if ir is not None else None
                        logits = _forward(model, r2, i2)[0:1]
                        _, c = ops.argmax_confusion(logits, label[b:b + 1], num_classes)
                        conf = c if conf is None else conf + c
                else:
                    logits = _forward(model, rgb, ir)
                    _, c = ops.argmax_confusion(logits, label, num_classes)
                    conf = c if conf is None else conf + c
    finally:
        model.train(was_training)
    conf = conf.cpu().numpy()
    ious = miou.ious_from_confusion(conf)
    return {"confusion": conf, "ious": ious, "miou": float(np.nanmean(ious)),
            "iou_eval": miou.iou_eval_value(conf, (12, 13))}
