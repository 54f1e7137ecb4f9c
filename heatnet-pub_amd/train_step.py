"""The HeatNet training iteration (losses + phase machine) on the HIP kernels.

Restates the body of the hot loop of models/confusion_maximization/train_trgb_segnet_conf.py:383-592 for the default
configuration (no night supervision, no cert branch, critic weights all 1): the reference trainer itself is a
module-level script that cannot be imported (wandb.init at import), so this module is what a caller drives.
"""
import torch

from . import ops


def critic_loss(out, adv="MSE"):
    """:437-446 -- critics learn day -> 1, night -> 0."""
    total = None
    for c in out["critics_a"]:
        term = ops.adv_loss(c, 1.0, adv)
        total = term if total is None else total + term
    for c in out["critics_b"]:
        total = total + ops.adv_loss(c, 0.0, adv)
    return total


def confusion_loss(out, adv="MSE", weights=None, multidir=False):
    """:526-540 -- the segmentation net is trained so that both domains look like 'day' (1) to every critic.

    The reference multiplies the mean-reduced criterion by a bilinear-resized all-ones map and averages again,
    which is the criterion itself.
    """
    n = len(out["critics_a"])
    weights = [1.0] * n if weights is None else weights
    total = None
    for m, c in enumerate(out["critics_a"]):
        term = ops.adv_loss(c, 0.0 if multidir else 1.0, adv) * weights[m]
        total = term if total is None else total + term
    for m, c in enumerate(out["critics_b"]):
        total = total + ops.adv_loss(c, 1.0, adv) * weights[m]
    return total


def segmentation_loss(logits, labels, ignore_index=-100):
    """nn.CrossEntropyLoss() of :244,452 (HeatNet) / ignore_index=13 of scripts/main.py:223."""
    return ops.cross_entropy(logits, labels, ignore_index)


class PhaseSchedule:
    """:361-367,577-592 -- start with `initial_critic` critic iterations, then alternate seg / critic phases."""

    def __init__(self, initial_critic=1000, seg=50, critic=500, no_conf=False):
        self.state = "train_seg" if no_conf else "train_critic"
        self.counter = initial_critic
        self.seg, self.critic, self.no_conf = seg, critic, no_conf

    def tick(self):
        if self.no_conf:
            return False
        self.counter -= 1
        if self.counter != 0:
            return False
        if self.state == "train_seg":
            self.state, self.counter = "train_critic", self.critic
        else:
            self.state, self.counter = "train_seg", self.seg
        return True


def certainty_confusion_weights(cert, out, critic_weights=None):
    """:531-540 with `--weight_ir_sup`: the confusion term of critic m is multiplied by
    mean(F.interpolate((1 - cert)[None], size=critic map size, mode="bilinear")) -- the reference multiplies the (scalar,
    mean-reduced) criterion by the resized map and averages, which is that product.  Returns one device scalar per critic.
    (1 - cert and the means are a handful of tiny stock-torch ops on an optional branch; the resize is hn_bilinear_fwd.)"""
    n_critics = len(out["critics_a"])
    critic_weights = [1.0] * n_critics if critic_weights is None else critic_weights
    one_minus = (1.0 - cert).unsqueeze(0)  # [1, N, H, W]: the reference resizes it with the batch as channels
    scalars, cache = [], {}
    for m, c in enumerate(out["critics_a"]):
        size = (c.size(2), c.size(3))
        if size not in cache:
            resized = one_minus if size == tuple(one_minus.shape[2:]) else ops.bilinear(one_minus, size)
            cache[size] = resized.mean()
        scalars.append(cache[size] * critic_weights[m])
    return scalars


def train_step(model, optimizer, day, night, label_day, state, adv="MSE", conf_weight=0.1, grad_sync=None,
               night_teacher=None, teacher_modalities=("ir",), weight_ir_sup=False):
    """One iteration: zero_grad, forward (day + night passes, critics), loss of the phase, backward, step.

    `model` is a `conv_segnet` (possibly wrapped so that `.module` exists); `grad_sync`, when given, is a
    `ddp.GradBuckets` whose all-reduce overlaps the backward pass.  Returns the loss tensors (no host sync).
    `night_teacher` (optional): the night-supervision model of train_trgb_segnet_conf.py:209-229 -- a frozen
    PSPNet fed the night thermal image (`teacher_modalities=("ir",)`) or RGB + thermal; in `train_seg` its arg-max
    prediction supervises the night pass (:430-434, 455-457).  `weight_ir_sup=True` is the trainer's `--weight_ir_sup`
    without a cert branch (PSPNet has none: :467-477, 531-532): the night cross-entropy is weighted per pixel by the
    teacher's certainty max(softmax(softmax(logits))) and averaged over the map, and every critic's confusion term is
    scaled by the mean of (1 - certainty) resized to that critic's map.
    """
    optimizer.zero_grad(set_to_none=grad_sync is None)  # flat-arena optimizers always zero in place
    if grad_sync is not None:
        grad_sync.begin_step()
    with ops.frozen_weights():  # day and night passes share one preparation of every conv's weight operands
        ops.prepare_weights()   # ... made by one launch for all layers seen in earlier iterations
        out = model(day, night)
    losses = {}
    if state == "train_seg":
        seg = segmentation_loss(out["pred_label_a"], label_day)
        conf_scalars = None
        if night_teacher is not None:
            with torch.no_grad():
                t_in = [night[0], night[1]] if "rgb" in teacher_modalities else [night[1]]
                t_logits = night_teacher(*t_in)[0]
                pseudo, cert = ops.teacher_outputs(t_logits, want_cert=weight_ir_sup)  # argmax(softmax(.)), certainty
            if weight_ir_sup:
                seg_night = ops.cross_entropy_weighted(out["pred_label_b"], pseudo, cert)
                if "critics_a" in out:
                    with torch.no_grad():
                        conf_scalars = certainty_confusion_weights(cert, out)
            else:
                seg_night = segmentation_loss(out["pred_label_b"], pseudo)
            losses["seg_night"] = seg_night.detach()
            seg = seg + seg_night
        inner = getattr(model, "module", model)
        if getattr(inner, "cert_branch", False) and night_teacher is None:
            # :482-491 (`--cert_branch`, no night supervision): the certainty head regresses 1 - p(true class) of the day pass
            if out.get("cert_a") is None:
                raise TypeError("cert_branch: the segmentation network returned no certainty map (only arch='custom' has the head)")
            cert = ops.cert_loss(out["pred_label_a"], label_day, out["cert_a"])
            losses["cert"] = cert.detach()
            seg = seg + cert
        losses["seg"] = seg.detach()
        if "critics_a" in out:
            conf = confusion_loss(out, adv, weights=conf_scalars)
            total = seg + conf_weight * conf
            losses["conf"] = conf.detach()
        else:
            total = seg
    else:
        total = critic_loss(out, adv)
    total.backward()
    if grad_sync is not None:
        grad_sync.finish_step()
    optimizer.step()
    ops.range_guard_tick()
    losses["total"] = total.detach()
    return losses


def segmentation_step(model, optimizer, rgb, ir, labels, ignore_index=-100, grad_sync=None):
    """BASELINE config 2: dual-stream PSPNet forward/backward + CE + optimizer step (no critics)."""
    with ops.main_stream(rgb.device):
        return _segmentation_step(model, optimizer, rgb, ir, labels, ignore_index, grad_sync)


def _segmentation_step(model, optimizer, rgb, ir, labels, ignore_index, grad_sync):
    optimizer.zero_grad(set_to_none=grad_sync is None)
    if grad_sync is not None:
        grad_sync.begin_step()
    with ops.frozen_weights():
        ops.prepare_weights()  # one launch prepares the weight operands of every conv seen in earlier iterations
        res = model(rgb, ir) if ir is not None else model(rgb)
    logits = res[0] if isinstance(res, tuple) else res
    loss = segmentation_loss(logits, labels, ignore_index)
    loss.backward()
    if grad_sync is not None:
        grad_sync.finish_step()
    optimizer.step()
    ops.range_guard_tick()  # HEATNET_F16X2_GUARD: layers with heavy-tailed operands leave the f16x2 arithmetic
    return loss.detach()
