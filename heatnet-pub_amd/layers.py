"""nn.Module building blocks whose arithmetic is the HIP kernels in `ops`.

They subclass the stock torch modules they replace and keep the same parameter names and logical shapes,
because the reference's callers re-initialise networks by type / class name (`isinstance(m, nn.Conv2d)` in
scripts/main.py:361-385; `'Conv'` / `'BatchNorm2d'` in the class name in
models/confusion_maximization/utils.py:126-132) and exchange `state_dict`s.
"""
import torch
import torch.nn as nn

from . import ops


def _square(v, what):
    if isinstance(v, (tuple, list)):
        if len(set(v)) != 1:
            raise ValueError("HnConv2d supports equal %s in both dimensions, got %r" % (what, v))
        return int(v[0])
    return int(v)


_DP_MESSAGE = ("heatnet_amd modules cannot be replicated by nn.DataParallel (one process, several GPUs, several threads): "
               "the kernels' tuning / weight-preparation caches and per-layer workspaces are per process and per device. "
               "Run one process per GPU -- torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank]) as "
               "scripts/main.py:123 does, or heatnet_amd.ddp.GradBuckets (see INTEGRATION.md).")


class _NoDataParallel:
    """nn.DataParallel over ONE device never replicates and keeps working; over several it is refused loudly instead of
    producing replicas that share device-0 workspaces."""

    def _replicate_for_data_parallel(self):
        raise RuntimeError(_DP_MESSAGE)


class HnConv2d(_NoDataParallel, nn.Conv2d):
    """nn.Conv2d on the implicit-GEMM MFMA kernels.  Weights live in channels-last (OHWI) memory."""

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        if self.groups != 1 or self.padding_mode != "zeros" or isinstance(self.padding, str):
            raise ValueError("HnConv2d: groups=1 and numeric zero padding only")
        self.weight.data = self.weight.data.contiguous(memory_format=torch.channels_last)

    def geometry(self):
        return (_square(self.stride, "stride"), _square(self.padding, "padding"), _square(self.dilation, "dilation"))

    def forward(self, x, act=ops.ACT_NONE, slope=0.0):
        s, p, d = self.geometry()
        return ops.conv2d(x, self.weight, self.bias, s, p, d, act, slope)


class HnBatchNorm2d(_NoDataParallel, nn.BatchNorm2d):
    """nn.BatchNorm2d (affine, tracked running stats) on the fused BN kernels."""

    _pending_batches = 0  # forward passes not yet added to `num_batches_tracked` (flushed on state_dict())

    def _flush_batches(self):
        if self._pending_batches and self.num_batches_tracked is not None:
            self.num_batches_tracked.add_(self._pending_batches)
        self._pending_batches = 0

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        self._flush_batches()
        super()._save_to_state_dict(destination, prefix, keep_vars)

    def _load_from_state_dict(self, *args, **kwargs):
        self._pending_batches = 0
        super()._load_from_state_dict(*args, **kwargs)

    def forward(self, x, residual=None, act=ops.ACT_NONE, stats=None, prelu_weight=None, drop_mask=None):
        training = self.training or self.running_mean is None
        if training and self.num_batches_tracked is not None:
            self._pending_batches += 1  # host-side counter: no 1-element kernel per layer per step
        momentum = 0.1 if self.momentum is None else self.momentum
        return ops.bn_act(x, stats, self.weight, self.bias, self.running_mean, self.running_var, residual,
                          prelu_weight, training, momentum, self.eps, act, ws=self.workspace(x.device),
                          drop_mask=drop_mask)

    _ws = None

    def workspace(self, device):
        """Per-layer double accumulators for batch statistics (forward) and the backward reductions."""
        if self._ws is None or self._ws.device != device:
            self._ws = ops.BnWorkspace(self.num_features, device)
        return self._ws


def conv_bn_act(x, conv, bn, act=ops.ACT_NONE, residual=None, prelu_weight=None, passthrough=False, drop_mask=None,
                out="f32", mask_max=1.0):
    """conv -> BatchNorm (batch statistics from the conv epilogue) -> (+residual) -> activation.

    `passthrough=True` additionally returns an alias of `x` to be used for the block's skip connection: its gradient
    then joins this conv's data gradient inside the dgrad kernel instead of in a separate elementwise add.
    `out`: "f32" (default), or -- a request the fused f16x2 node honours where it covers the layer, ignored elsewhere --
    "planes" (the result as a plane tensor: ONLY for tensors whose every consumer is another conv_bn_act / its residual
    input) or "dual" (fp32 carrying a plane twin).  `x` and `residual` may be plane tensors."""
    s, p, d = conv.geometry()
    batch_stats = bn.training or bn.running_mean is None
    if ops.fused_conv_bn_ok(x, conv.weight, s, p, d, batch_stats):
        if bn.training and bn.num_batches_tracked is not None:
            bn._pending_batches += 1
        momentum = 0.1 if bn.momentum is None else bn.momentum
        leaky = 0.0
        return ops.conv_bn_act_fused(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean, bn.running_var, residual,
                                     prelu_weight, s, p, d, act, leaky, momentum, bn.eps, bn.workspace(x.device), drop_mask,
                                     passthrough, out, mask_max)
    x, residual = ops.as_f32(x), ops.as_f32(residual)
    ws = bn.workspace(x.device)
    stats = ws.stats if bn.training else None  # conv epilogue adds into it, hn_bn_finalize re-zeroes
    if stats is not None:
        ws.open_stats()
    out_ = ops.conv2d(x, conv.weight, conv.bias, s, p, d, want_stats=stats if stats is not None else False,
                      passthrough=passthrough, zero_bias_grad=batch_stats and conv.bias is not None)
    y, skip = out_ if passthrough else (out_, None)
    z = bn(y, residual=residual, act=act, stats=stats, prelu_weight=prelu_weight, drop_mask=drop_mask)
    return (z, skip) if passthrough else z


class HnPReLU(nn.PReLU):
    """Single-slope nn.PReLU.  Inside PSPUpsample it is fused behind the BatchNorm; standalone it is an
    identity-affine BN-free activation pass."""

    def forward(self, x):
        raise RuntimeError("HnPReLU is fused into the preceding BatchNorm kernel; call the parent block instead")


class HnDropout2d(nn.Dropout2d):
    def forward(self, x, mask=None):
        return ops.dropout2d(x, self.p, self.training, mask)

    def mask_for(self, n, c, device):
        """The [N, C] keep/(1-p) factors this layer would apply now (None when it is the identity), for producers
        that fold the multiplication into their own output pass."""
        if not self.training or self.p == 0.0:
            return None
        return ops.drop_mask(n, c, self.p, device)


class HnMaxPool3x3s2(nn.Module):
    """nn.MaxPool2d(kernel_size=3, stride=2, padding=1)."""

    def forward(self, x):
        return ops.maxpool3x3s2(x)


class HnAdaptiveAvgPool2d(nn.Module):
    def __init__(self, output_size):
        super().__init__()
        self.output_size = output_size

    def forward(self, x):
        return ops.adaptive_avg_pool(x, self.output_size)


def ensure_channels_last_(module):
    """Re-lay conv weights as OHWI after something (a checkpoint load, `.to()`) made them NCHW-contiguous."""
    for m in module.modules():
        if isinstance(m, nn.Conv2d) and m.weight.dim() == 4:
            w = m.weight.data
            if not w.is_contiguous(memory_format=torch.channels_last):
                m.weight.data = w.contiguous(memory_format=torch.channels_last)
    return module
