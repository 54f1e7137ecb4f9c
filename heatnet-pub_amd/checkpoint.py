"""Checkpoint interchange with the reference's formats.

    models/confusion_maximization/utils.py:59-90      initModelRenamed / initModelPartial / initModelFull
    models/confusion_maximization/train_trgb_segnet_conf.py:112-116,648-654   {'epoch','state_dict','best_iou','optimizer','lr_scheduler'}
Tensors are stored in logical OIHW / (C,) shapes, so files move both ways; keys may carry the `module.` prefix of
nn.DataParallel.
"""
import shutil

import torch

from .layers import ensure_channels_last_


def _state(obj):
    return obj["state_dict"] if isinstance(obj, dict) and "state_dict" in obj else obj


def init_model_renamed(model, weights_path, to_rename, rename):
    saved = _state(torch.load(weights_path, map_location="cpu"))
    own = model.state_dict()
    picked = {k.replace(to_rename, rename): v for k, v in saved.items()}
    picked = {k: v for k, v in picked.items() if k in own}
    print("Loaded dict with %d entries..." % len(picked))
    assert len(picked) > 0
    own.update(picked)
    model.load_state_dict(own)
    return ensure_channels_last_(model)


def init_model_partial(model, weights_path):
    saved = torch.load(weights_path, map_location="cpu")["state_dict"]
    own = model.state_dict()
    own.update({k: v for k, v in saved.items() if k in own})
    model.load_state_dict(own)
    return ensure_channels_last_(model)


def init_model_full(model, weights_path):
    model.load_state_dict(torch.load(weights_path, map_location="cpu"))
    return ensure_channels_last_(model)


def save_checkpoint(state, is_best, filename="checkpoint.pth.tar"):
    """torch.save + copy to `<filename>_best.pth.tar` (train_trgb_segnet_conf.py:112-116, minus wandb.save)."""
    sd = {k: v.contiguous() if torch.is_tensor(v) else v for k, v in state.get("state_dict", {}).items()}
    state = dict(state, state_dict=sd)
    torch.save(state, filename)
    if is_best:
        shutil.copyfile(filename, filename + "_best.pth.tar")


def resume(path, model, optimizer=None, lr_scheduler=None):
    """The resume block of cm/train_trgb_segnet_conf.py:276-283: model weights, optimizer and LR-scheduler state, epoch
    and best mIoU from a `save_checkpoint` file of either implementation.  Returns (epoch, best_iou).  `optimizer` may be
    a torch.optim optimizer or one of optim.FlatRMSprop / FlatAdam (state is matched by parameter order)."""
    ckpt = torch.load(path, map_location="cpu")
    state = ckpt["state_dict"]
    own = model.state_dict()
    if not any(k in own for k in state) and all(k.startswith("module.") for k in state):
        state = {k[len("module."):]: v for k, v in state.items()}  # written through nn.DataParallel
    model.load_state_dict(state)
    ensure_channels_last_(model)
    if optimizer is not None and "optimizer" in ckpt:
        optimizer.load_state_dict(ckpt["optimizer"])
    if lr_scheduler is not None and "lr_scheduler" in ckpt:
        lr_scheduler.load_state_dict(ckpt["lr_scheduler"])
    return ckpt.get("epoch", 0), ckpt.get("best_iou", 0.0)
