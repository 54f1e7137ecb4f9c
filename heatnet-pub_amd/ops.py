"""Operators of the HeatNet hot path: `torch.autograd.Function`s whose forward and backward are the
hand-written gfx950 kernels of libheatnet_hip.so, called through the C ABI (see `_lib.py`).

Tensors keep the reference's LOGICAL NCHW shape (so callers, `state_dict`s and losses see what the
reference produces) but are physically NHWC: a 4-D tensor here has stride(1) == 1 and a pixel stride
`ld >= C`, i.e. torch's channels-last format or a channel slice of such a buffer.  PyTorch is used for
allocation, autograd bookkeeping and streams only -- no arithmetic of the hot path runs in ATen.
"""
import ctypes
import os
import sys
import weakref

import torch

from . import _lib
from ._lib import HnConvDesc, HnWeightPrepItem, check

ACT_NONE, ACT_RELU, ACT_PRELU, ACT_LEAKY = 0, 1, 2, 3
STAT_REPLICAS = 16  # HN_STAT_REPLICAS in include/heatnet_hip.h
LOSS_MSE, LOSS_BCE = 0, 1


# ------------------------------------------------------------------------------------------------------
# tensor plumbing
# ------------------------------------------------------------------------------------------------------

def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _require_gpu(t):
    if getattr(t, "_hn_planes", None) is not None:
        raise RuntimeError("heatnet_amd: a plane tensor (fp16 hi / lo images, see PlaneTensor in ops.py) reached an operator "
                           "that reads fp32; only conv_bn_act / the residual input of bn take that format")
    if not t.is_cuda:
        raise RuntimeError("heatnet_amd operators run on the MI355X only (got a %s tensor); there is no CPU fallback"
                           % t.device)
    if t.dtype != torch.float32:
        raise RuntimeError("heatnet_amd operators compute in fp32, got %s" % t.dtype)


def nhwc_ld(t):
    """Pixel stride of `t` if it is NHWC-strided (see module docstring), else None."""
    n, c, h, w = t.shape
    if c > 1 and t.stride(1) != 1:
        return None
    if w > 1:
        ld = t.stride(3)
    elif h > 1:
        ld = t.stride(2)
    elif n > 1:
        ld = t.stride(0)
    else:
        ld = c
    if ld < c:
        return None
    if (w > 1 and t.stride(3) != ld) or (h > 1 and t.stride(2) != w * ld) or (n > 1 and t.stride(0) != h * w * ld):
        return None
    return ld


def as_nhwc(t):
    """Return (tensor, ld) with NHWC strides, copying only when `t` is laid out otherwise."""
    _require_gpu(t)
    ld = nhwc_ld(t)
    if ld is None:
        n, c, h, w = t.shape
        if t.is_contiguous() and not (torch.is_grad_enabled() and t.requires_grad):  # plain NCHW: our own transposing kernel
            out = new_nhwc(n, c, h, w, t.device)
            check(_lib.load().hn_nchw_to_nhwc(_ptr(t), _ptr(out), n, c, h * w, c, _stream()), "hn_nchw_to_nhwc")
            return out, c
        t = t.contiguous(memory_format=torch.channels_last)
        ld = nhwc_ld(t)
        if ld is None:  # degenerate sizes for which torch reports ambiguous strides
            t = t.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
            ld = c
    return t, ld


def new_nhwc(n, c, h, w, device, dtype=torch.float32):
    return torch.empty((n, h, w, c), device=device, dtype=dtype).permute(0, 3, 1, 2)


def _vector_grad(dz, lddz):
    """The byte ReLU masks (and the bf16 y) exist for whole 16-byte pieces only: a gradient that autograd hands over as an
    unaligned view, or with a pixel stride that is no multiple of four, is re-laid out dense (the saved z is gone, so the scalar
    kernels are no alternative)."""
    if dz.data_ptr() % 16 == 0 and lddz % 4 == 0:
        return dz, lddz
    n, c, h, w = dz.shape
    out = new_nhwc(n, c, h, w, dz.device)
    out.copy_(dz)
    return out, c


def weight_ohwi(w):
    """Conv weight (logical OIHW) as a dense [O][KH][KW][I] buffer; a no-op for channels-last parameters."""
    _require_gpu(w)
    o, i, kh, kw = w.shape
    expect = (kh * kw * i, 1, kw * i, i)
    ok = all(w.shape[d] == 1 or w.stride(d) == expect[d] for d in range(4))
    if not ok:
        if kh == 1 and kw == 1 and w.stride(1) == 1 and w.stride(0) >= i and not (torch.is_grad_enabled() and w.requires_grad):
            # an input-channel slice of a wider 1x1 weight (the pyramid bottleneck's columns): rows of i floats, stride ld
            out = torch.empty((o, 1, 1, i), device=w.device, dtype=w.dtype).permute(0, 3, 1, 2)
            check(_lib.load().hn_copy(_ptr(w), w.stride(0), _ptr(out), i, o, i, _stream()), "hn_copy")
            return out
        w = w.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    return w


def conv_out_size(h, k, stride, pad, dil):
    return (h + 2 * pad - dil * (k - 1) - 1) // stride + 1


def _desc(n, h, w, cin, ldx, ho, wo, cout, ldy, kh, kw, stride, pad, dil):
    return HnConvDesc(n, h, w, cin, ldx, ho, wo, cout, ldy, kh, kw, stride, pad, dil)


# ---- per-shape kernel-variant selection ---------------------------------------------------------------------------
# hn_conv2d_* take a `variant` (tile sides, K step) whose results are bitwise identical; which one is fastest depends
# on how the grid divides over the 256 CUs.  On first use of a geometry every candidate is timed once with HIP events
# and the winner is cached.  HEATNET_AUTOTUNE=0 keeps the library's built-in heuristics.
AUTOTUNE = os.environ.get("HEATNET_AUTOTUNE", "1") != "0"
_variant_cache = {}
# Choices persist across processes in a small JSON file (geometry -> variant) so that later runs, and profiled runs,
# contain no candidate-timing launches.  HEATNET_TUNE_CACHE overrides the location; an unreadable / unwritable file is
# simply ignored.  Only rank 0 writes.
_TUNE_FILE = os.environ.get("HEATNET_TUNE_CACHE") or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                                  "tune_cache.json")
_tune_dirty = False


def _load_tune_cache():
    import json
    try:
        with open(_TUNE_FILE) as f:
            data = json.load(f)
        if data.pop("_version", None) != _lib.load().hn_version():
            return  # timings of another kernel generation
        for k, v in data.items():
            if True:
                key = tuple(int(x) if x.lstrip("-").isdigit() else x for x in k.split(","))
                _variant_cache[key] = tuple(v) if isinstance(v, list) else v
    except (OSError, ValueError, RuntimeError):
        pass


def _save_tune_cache():
    import json
    if not _tune_dirty or os.environ.get("RANK", "0") != "0":
        return
    try:
        data = {",".join(str(x) for x in k): (list(v) if isinstance(v, tuple) else v) for k, v in _variant_cache.items()}
        data["_version"] = _lib.load().hn_version()
        tmp = _TUNE_FILE + ".tmp.%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(data, f, indent=0, sort_keys=True)
        os.replace(tmp, _TUNE_FILE)
    except OSError:
        pass


if AUTOTUNE:
    import atexit
    _load_tune_cache()
    atexit.register(_save_tune_cache)


# ---- arithmetic of the convolution kernels --------------------------------------------------------------------------
# "f32":    v_mfma_f32_32x32x2_f32 on the fp32 operands.
# "bf16x3": every fp32 operand split exactly into three bf16 terms in the kernel, six bf16 MFMAs per product, fp32
#           accumulate: the same fp32 result quality (see conv_igemm.hip, profiles/r01_bf16x3_probe.txt) at the bf16
#           matrix-core rate / 6 instead of the fp32 one.  Shapes the bf16x3 kernels do not cover (Cin % 32 != 0) run
#           on the f32 kernels either way.
# "bf16":   REDUCED precision (BASELINE configs[4]): operands rounded to nearest bf16 as they are staged, one bf16 MFMA
#           per product, fp32 accumulation; tensors in memory, BatchNorm statistics, master weights and the optimizer
#           stay fp32.  Never the default; bench.py reports it as a separate workload with dtype "bf16".
# "f16x2":  fp32 results from TWO fp16 terms per operand: every operand tensor is scaled by a power of two (its largest
#           magnitude -> [2^14, 2^15)) and split into hi + lo fp16 terms, three products per fp32 product, fp32 accumulate
#           (include/heatnet_hip.h).  As accurate as the fp32 MFMA against fp64 where the reduction is long (the 2^-23
#           representation error of a product is below the accumulation error); layers whose reduction is shorter than
#           H2_MIN_K terms run on bf16x3, shapes neither covers on f32.
_MATH_CODES = {"f32": 1, "bf16x3": 2, "bf16": 3, "f16x2": 4}
_PLANE_MATHS = ("bf16x3", "bf16", "f16x2")  # arithmetics that take the weight operand as pre-made 16-bit planes
H2_MIN_K = int(os.environ.get("HEATNET_F16X2_MIN_K", "0"))
CONV_MATH = os.environ.get("HEATNET_CONV_MATH", "f16x2")
if CONV_MATH not in _MATH_CODES:
    raise RuntimeError("HEATNET_CONV_MATH must be one of %s" % sorted(_MATH_CODES))
# diagnostics only: comma-separated launch kinds (fwd, dgrad, wgrad) kept on the fp32 MFMA while CONV_MATH is bf16x3
_MATH_F32_KINDS = set(filter(None, os.environ.get("HEATNET_CONV_MATH_F32_KINDS", "").split(",")))


def _math(kind):
    return "f32" if kind in _MATH_F32_KINDS else CONV_MATH


def _layer_math(kind, cin, cout, kh, kw, leaf=None):
    """Arithmetic of one launch kind of one layer: "f16x2" only where both reductions of the layer (forward: KH*KW*Cin,
    data gradient: KH*KW*Cout) have at least H2_MIN_K terms, else "bf16x3" -- one format of weight planes per layer.
    `leaf`: the layer's weight parameter -- a layer the range guard moved off f16x2 (`_hn_no_f16x2`) runs bf16x3."""
    m = _math(kind)
    if m == "f16x2" and (kh * kw * min(cin, cout) < H2_MIN_K or cin % 32 != 0 or cin * cout * kh * kw < 256 or
                         getattr(leaf, "_hn_no_f16x2", False)):
        return "bf16x3"
    return m


# ---- f16x2 range guard ---------------------------------------------------------------------------------------------------
# f16x2 keeps 22 significant bits of an operand element only if it lies within 2^-18 of its tensor's magnitude bound; a
# tensor with a heavy tail (a few huge values over many tiny, non-zero ones) loses bits on the small ones, bf16x3 does not.
# HEATNET_F16X2_GUARD=<fraction> (e.g. 0.05) turns on a census of every f16x2 operand (hn_range_census: non-zero elements,
# and those below bound * 2^-18) -- one extra pass over each operand, a diagnostic / safety mode, not the fast path -- and
# `range_guard_update()` (called by train_step every GUARD_EVERY iterations) moves every layer whose activation or gradient
# operand had more than that fraction of its non-zero elements in the lossy range over to bf16x3, for good.
# HEATNET_F16X2_CENSUS=1 only counts (tools/trained_state_f16x2.py); `range_report()` reads the counters.
RANGE_GUARD = float(os.environ.get("HEATNET_F16X2_GUARD", "0") or 0.0)
RANGE_CENSUS = RANGE_GUARD > 0 or os.environ.get("HEATNET_F16X2_CENSUS", "0") == "1"
GUARD_EVERY = int(os.environ.get("HEATNET_F16X2_GUARD_EVERY", "25"))
_census = {}  # id(weight leaf) -> [weakref, geometry, counters (int64 [2 operands][nonzero, low])]
guard_moved = []  # (geometry, fraction activations, fraction gradients) of the layers moved to bf16x3 so far


def set_range_census(on):
    global RANGE_CENSUS
    prev, RANGE_CENSUS = RANGE_CENSUS, bool(on) or RANGE_GUARD > 0
    return prev


def _census_operand(leaf, which, x, planes, ld, slot, npix, c):
    """Count operand `which` (0: the activation operand of forward / wgrad, 1: the gradient operand of dgrad / wgrad)."""
    if leaf is None:
        return
    e = _census.get(id(leaf))
    if e is None or e[0]() is not leaf:
        e = _census[id(leaf)] = [weakref.ref(leaf), tuple(leaf.shape), torch.zeros(4, device=slot.device, dtype=torch.int64)]
    check(_lib.load().hn_range_census(_ptr(x), _ptr(planes), npix * ld if planes is not None else 0, ld, _ptr(slot), npix, c,
                                      ctypes.c_void_p(e[2].data_ptr() + 16 * which), _stream()), "hn_range_census")


def range_report(reset=False):
    """[(weight shape, non-zero activations, fraction of them below bound * 2^-18, non-zero gradient elements, fraction)] per
    f16x2 layer counted since the last reset.  Synchronises."""
    rows = []
    for key in list(_census):
        ref, geom, cnt = _census[key]
        leaf = ref()
        if leaf is None:
            del _census[key]
            continue
        v = cnt.tolist()
        rows.append((geom, v[0], v[1] / v[0] if v[0] else 0.0, v[2], v[3] / v[2] if v[2] else 0.0, leaf))
        if reset:
            cnt.zero_()
    return rows


def range_guard_update(threshold=None):
    """Move every layer with more than `threshold` (default HEATNET_F16X2_GUARD) of an operand's non-zero elements in f16x2's
    lossy range to bf16x3; returns the layers moved by this call.  Counters start again."""
    threshold = RANGE_GUARD if threshold is None else threshold
    moved = []
    for geom, _, fa, _, fg, leaf in range_report(reset=True):
        if max(fa, fg) > threshold and not getattr(leaf, "_hn_no_f16x2", False):
            leaf._hn_no_f16x2 = True
            leaf._hn_prep = None  # weight planes of the other format are made on next use
            moved.append((geom, fa, fg))
    guard_moved.extend(moved)
    return moved


_guard_calls = 0


def range_guard_tick():
    """Called once per training iteration by train_step: every GUARD_EVERY calls the counters are read and layers moved."""
    global _guard_calls
    if RANGE_GUARD <= 0:
        return
    _guard_calls += 1
    if _guard_calls % GUARD_EVERY == 0:
        for geom, fa, fg in range_guard_update():
            print("heatnet_amd: f16x2 range guard: conv %s -> bf16x3 (%.1f%% of the activation operand, %.1f%% of the gradient "
                  "operand below bound * 2^-18)" % (geom, 100 * fa, 100 * fg), file=sys.stderr)


def set_conv_math(name):
    """Select the conv arithmetic ("f32" | "bf16x3" | "f16x2" | "bf16") for subsequent launches; returns the previous setting."""
    global CONV_MATH
    if name not in _MATH_CODES:
        raise ValueError("conv math must be one of %s" % sorted(_MATH_CODES))
    prev, CONV_MATH = CONV_MATH, name
    return prev


# Deterministic mode (HEATNET_DETERMINISTIC=1 or set_deterministic(True)): no float atomics in the convolution kernels --
# forward / data-gradient launches do not use split-K, weight gradients go through hn_conv2d_wgrad_det (partial sums
# stored per reduction split, added in a fixed order).  Results are then bitwise reproducible run to run; it costs speed
# on the small-pixel-count layers and a workspace of (splits x dW) floats.  The double-precision BatchNorm sums keep
# their atomics (order effects of ~1e-16 relative, below fp32 resolution).
DETERMINISTIC = False
_det_ws = {}


def set_deterministic(on):
    global DETERMINISTIC
    prev, DETERMINISTIC = DETERMINISTIC, bool(on)
    check(_lib.load().hn_set_deterministic(int(DETERMINISTIC)), "hn_set_deterministic")
    return prev


def _det_workspace(device, numel):
    # one per STREAM: the two encoder streams of a late-fusion network (resnet.ResNet.forward) run weight gradients concurrently
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _det_ws.get(key)
    if buf is None or buf.numel() < numel:
        buf = _det_ws[key] = torch.empty(numel, device=device, dtype=torch.float32)
    return buf


def _variant(bk=0, tm=0, tn=0, math=0):
    return bk + 4 * tm + 16 * tn + 64 * math


# Prepared weight operands (bf16 planes, flipped / transposed copies) are reused only inside a `frozen_weights()` scope:
# the caller's promise that no parameter changes while it is open (the day and night passes of one training iteration,
# an evaluation loop).  Outside a scope every forward prepares its operands afresh -- parameter writes through `.data`
# (the reference's init functions) cannot be observed, so nothing is ever reused on a guess.
_frozen_depth = 0
_frozen_scope = 0


class frozen_weights:
    """`with ops.frozen_weights(): ...` -- parameters are constant inside: weight operands prepared by one forward are
    reused by the following ones (keyed on the parameter's storage, version counter and conv geometry as well)."""

    def __enter__(self):
        global _frozen_depth, _frozen_scope
        if _frozen_depth == 0:
            _frozen_scope += 1
            _prep_registry.new_scope()
        _frozen_depth += 1
        return self

    def __exit__(self, *exc):
        global _frozen_depth
        _frozen_depth -= 1
        return False


def weights_changed():
    """Invalidate operands prepared so far (a new scope id); for code that updates parameters inside an open
    `frozen_weights()` scope after all."""
    global _frozen_scope
    _frozen_scope += 1


class _WeightPrepRegistry:
    """The convolutions seen inside frozen_weights() scopes (parameter, geometry, persistent operand buffers).  Once a
    model's layers are known, `prepare_weights()` refreshes the operands of all layers used in the PREVIOUS scope with one
    launch (hn_weight_prepare_batch) instead of one small launch per layer and step.  Parameters are held weakly."""

    def __init__(self):
        self.entries = {}      # (id(param), stride, pad, dil) -> dict(ref, geom, wx3, wt, wtx3, last_scope)
        self.signature = None  # what the device table was built for
        self.table = None
        self.n = self.blocks = self.has_h2 = 0
        # scope ids are not consecutive (weights_changed() -- every optimizer step -- bumps them too): remember which
        # id the layers of the previous frozen_weights() scope were touched under
        self.prev_touch = self.cur_touch = None

    def new_scope(self):
        if self.cur_touch is not None:
            self.prev_touch, self.cur_touch = self.cur_touch, None

    def touch(self, leaf, geom, wx3, wt, wtx3, fmt=0):
        key = (id(leaf),) + geom
        e = self.entries.get(key)
        if e is None or e["ref"]() is not leaf or e["wx3"] is not wx3:
            e = self.entries[key] = dict(ref=weakref.ref(leaf), geom=geom, wx3=wx3, wt=wt, wtx3=wtx3, fmt=fmt)
        e["last_scope"] = self.cur_touch = _frozen_scope

    def run(self):
        dead = [k for k, e in self.entries.items() if e["ref"]() is None]
        for k in dead:
            del self.entries[k]
        if self.prev_touch is None:
            return
        live = [(e["ref"](), e) for e in self.entries.values() if e["last_scope"] == self.prev_touch]
        live = [(w, e) for w, e in live if w is not None and w.is_cuda]
        if not live:
            return
        lib = _lib.load()
        sig = tuple((w.data_ptr(), e["geom"], e["wx3"].data_ptr(), e["wt"].data_ptr() if e["wt"] is not None else 0,
                     e["wtx3"].data_ptr() if e["wtx3"] is not None else 0, e["fmt"]) for w, e in live)
        if sig != self.signature:  # first use, parameters re-homed (arena), or another set of layers (phase switch)
            items = (HnWeightPrepItem * len(live))()
            for it, (w, e) in zip(items, live):
                cout, cin, kh, kw = w.shape
                stride, pad, dil = e["geom"]
                it.w, it.w_x3 = w.data_ptr(), e["wx3"].data_ptr()
                it.wt = e["wt"].data_ptr() if e["wt"] is not None else None      # forward-only layers (frozen phase):
                it.wt_x3 = e["wtx3"].data_ptr() if e["wtx3"] is not None else None  # planes of w only
                it.Cout, it.KH, it.KW, it.Cin, it.stride, it.pad, it.dil = cout, kh, kw, cin, stride, pad, dil
                it.format = e["fmt"]
            host = torch.empty(lib.hn_weight_prep_table_bytes(len(live)), dtype=torch.uint8)
            blocks = ctypes.c_int(0)
            check(lib.hn_weight_prep_table_fill(items, len(live), ctypes.c_void_p(host.data_ptr()), ctypes.byref(blocks)),
                  "hn_weight_prep_table_fill")
            self.table = host.to(live[0][0].device)
            self.n, self.blocks, self.signature = len(live), blocks.value, sig
            self.has_h2 = int(any(e["fmt"] == 1 for _, e in live))
        check(lib.hn_weight_prepare_batch(_ptr(self.table), self.n, self.blocks, self.has_h2, _stream()),
              "hn_weight_prepare_batch")
        for w, e in live:
            w._hn_prep = ((w.data_ptr(), w._version, _frozen_scope) + e["geom"] + (e["fmt"],), e["wx3"], e["wt"], e["wtx3"])
            e["last_scope"] = self.cur_touch = _frozen_scope


_prep_registry = _WeightPrepRegistry()
BATCH_PREP = os.environ.get("HEATNET_BATCH_PREP", "1") != "0"  # A/B knob


def prepare_weights():
    """Inside a frozen_weights() scope, before the forward passes: refresh the weight operands of every convolution seen
    in earlier scopes with ONE launch.  Layers not seen yet prepare themselves on first use (and are included next time).
    A no-op under the "f32" arithmetic or outside a scope."""
    if _frozen_depth > 0 and CONV_MATH in _PLANE_MATHS and not _MATH_F32_KINDS and BATCH_PREP:
        _prep_registry.run()


def x3_plane(n):
    """HN_X3_PLANE: elements per bf16 plane of a pre-split matrix of n elements."""
    return (n + 7) & ~7


def _kind(kind, math=None):
    """Tune-cache key of a launch kind under an arithmetic (default: the current one)."""
    math = math or _math(kind)
    return kind if math == "f32" else kind + "." + math


# ---- f16x2 operand scales ---------------------------------------------------------------------------------------------
# One slot of 64 32-bit words (on 64 different lines: SCALE_SLOT_FLOATS floats) per tensor (bit patterns of its largest magnitude, accumulated by the producing pass or by
# hn_absmax_scale; include/heatnet_hip.h), handed out from zero-filled chunks (one fill per 512 scales) and never reused; a
# chunk is released by the allocator when the last tensor / autograd node holding one of its slots is gone.  A scale is
# remembered on the tensor it was computed for (`_hn_scale`, with the tensor's version counter).
_ABSMAX_TRACE = os.environ.get("HEATNET_ABSMAX_TRACE") == "1"  # diagnostics: who still needs a pass of its own
SCALE_SLOT_FLOATS = 64 * 32  # HN_SCALE_SLOT_FLOATS: 64 words on 64 different 128-B lines
_scale_chunks = {}
absmax_launches = 0  # hn_absmax_scale passes made so far (diagnostics: scales not supplied by a producing kernel)


def _new_scale_slot(device):
    # chunks are per STREAM: a chunk is zero-filled on the stream that is current when it is made, and its slots must not be
    # written by another stream's kernels before that fill has run
    key = (device, torch.cuda.current_stream(device).cuda_stream) if device.type == "cuda" else device
    c = _scale_chunks.get(key)
    if c is None or c[1] >= c[0].shape[0]:
        c = _scale_chunks[key] = [torch.zeros((512, SCALE_SLOT_FLOATS), device=device, dtype=torch.float32), 0]
    slot = c[0][c[1]]
    c[1] += 1
    return slot


def tensor_scale(t, ld=None, owner=None):
    """f16x2 scale of the NHWC tensor `t` (pixel stride `ld`); `owner`: the tensor object to remember it on (the caller's
    original when `t` is a re-laid-out copy)."""
    owner = t if owner is None else owner
    hit = getattr(owner, "_hn_scale", None)
    if hit is not None and hit[0] == owner._version and hit[1].device == t.device:
        return hit[1]
    global absmax_launches
    absmax_launches += 1
    if _ABSMAX_TRACE:
        import traceback
        print("absmax", tuple(t.shape), [f.name for f in traceback.extract_stack()[-6:-1]], file=sys.stderr)
    n, c, h, w = t.shape
    slot = _new_scale_slot(t.device)
    check(_lib.load().hn_absmax_scale(_ptr(t), c if ld is None else ld, n * h * w, c, _ptr(slot), _stream()), "hn_absmax_scale")
    owner._hn_scale = (owner._version, slot)
    return slot


def _inherit_scale(out, src):
    """`out` holds no magnitude larger than `src` does (an alias, a max / average pooling, a bilinear resampling -- convex
    combinations of src's elements): src's scale slot is a valid, at most slightly conservative, bound for it."""
    hit = getattr(src, "_hn_scale", None)
    if hit is not None and hit[0] == src._version:
        out._hn_scale = (out._version, hit[1])
    return out


_h2_cover = {}


def _h2_covers(kind_idx, d):
    """Do the f16x2 kernels cover this forward (0) / data-gradient (1) geometry?"""
    key = (kind_idx,) + tuple(getattr(d, f) for f, _ in HnConvDesc._fields_)
    r = _h2_cover.get(key)
    if r is None:
        buf = ctypes.create_string_buffer(128)
        r = _h2_cover[key] = _lib.load().hn_conv2d_plan(kind_idx, ctypes.byref(d), 1, _variant(0, 0, 0, 4), buf, 128) >= 0
    return r


TUNE_FLUSH = os.environ.get("HEATNET_TUNE_FLUSH", "1") != "0"  # 0: time candidates back to back (cache-resident operands), see _pick
_flush = {}


def _current_device():
    return torch.device("cuda", torch.cuda.current_device())


def _flush_buffer(device):
    buf = _flush.get(device)
    if buf is None:
        buf = _flush[device] = torch.zeros(192 * 1024 * 1024, device=device, dtype=torch.float32)  # 768 MB > Infinity Cache + L2
    return buf


_candidates_seen = None  # a dict while tools/insitu_tune.py records the candidate lists


def _pick(kind, d, candidates, launch):
    if not AUTOTUNE or DETERMINISTIC:  # (deterministic mode changes the split-K plans the cached timings were made for)
        return candidates[0]
    key = (kind,) + tuple(getattr(d, f) for f, _ in HnConvDesc._fields_)
    if _candidates_seen is not None:
        _candidates_seen[key] = list(candidates)  # (tools/insitu_tune.py: which alternatives a geometry has)
    hit = _variant_cache.get(key)
    if hit is not None:
        return hit
    for cand in candidates:
        launch(cand)  # warm-up (first launch of an instantiation also sets its LDS attribute)
    times = {c: float("inf") for c in candidates}
    for _ in range(4):  # interleaved rounds, best-of: robust against clock ramps and one-off hiccups
        for cand in candidates:
            if TUNE_FLUSH:
                # operands of up to ~250 MB stay in the Infinity Cache between repeated launches, inside a step they come from
                # HBM: evict them, so that the HBM-bound short-reduction layers are timed as they will run
                _flush_buffer(_current_device()).add_(1.0)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            launch(cand)
            e1.record()
            e1.synchronize()
            times[cand] = min(times[cand], e0.elapsed_time(e1))
    # keep the library's heuristic choice (first candidate) unless another one is clearly faster
    best = min(candidates, key=lambda c: times[c])
    if times[best] > 0.97 * times[candidates[0]]:
        best = candidates[0]
    global _tune_dirty
    _variant_cache[key] = best
    _tune_dirty = True
    return best


_scratch = {}


def _tune_scratch(like):
    """Same-size scratch for candidate timing: those launches must not leave anything in a gradient slot."""
    buf = _scratch.get((like.device, like.numel(), like.dtype))
    if buf is None:
        buf = _scratch[(like.device, like.numel(), like.dtype)] = torch.empty(like.numel(), device=like.device, dtype=like.dtype)
    return buf


# bf16x3 schedules offered to the tuner (HEATNET_X3_SCHEDULES=1,3 adds the LDS-DMA one: measured 140.7 / 141.0 images/s with
# it against 141.2 / 141.5 without, profiles/r02_conv_schedules.txt run G -- so only the first by default)
_X3_SCHEDULES = tuple(int(v) for v in os.environ.get("HEATNET_X3_SCHEDULES", "1").split(","))


# f16x2 schedules offered to the tuner: 0 = two-phase (conv_igemm.hip); 3 = weights by LDS-DMA (conv_igemm3.hip: the 128 x 128
# tile with two workgroups per CU and the eight-wave 128 x 256 tile; +4-16 % on the large shapes, run K of
# profiles/r02_conv_schedules.txt); 2 = wave-specialised (conv_igemm2.hip: 0-30 % slower on every shape, run I -- only with
# HEATNET_H2_SCHEDULES=0,2,3).  All bitwise identical.
_H2_SCHEDULES = tuple(int(v) for v in os.environ.get("HEATNET_H2_SCHEDULES", "0,3").split(","))


def _igemm_candidates(cout, kind, cred=0, math_name=None):
    """Variant codes to time for an implicit-GEMM launch with `cout` output and `cred` reduction channels."""
    tiles = [(0, 0), (1, 2), (2, 1)] if cout > 64 else [(0, 0), (1, 1)]
    math_name = math_name or _math(kind)
    math = _MATH_CODES[math_name]
    # the `bk` field: K step 32 / 16 of the fp32-MFMA kernels; schedule of the bf16x3 kernels (1 = two-phase,
    # conv_igemm.hip; 2 = wave-specialised, conv_igemm2.hip -- slower on every shape of this network, so it is not a
    # tuning candidate; 3 = 128 x 256 tile with the weight planes by LDS-DMA, conv_igemm3.hip -- a few per cent faster on
    # some of the 3x3 and wide 1x1 layers in isolation, slower on others, no gain over a step); the bf16 kernels have one
    bks = {"f32": (1, 2), "bf16x3": (1,), "bf16": (0,), "f16x2": tuple(b for b in _H2_SCHEDULES if b != 3)}[math_name]
    cands = [_variant(bk, tm, tn, math) for tm, tn in tiles for bk in bks]
    if math_name in ("f16x2", "bf16") and 3 in _H2_SCHEDULES and cred % 32 == 0 and cred > 0:
        if cout >= 96:
            cands.append(_variant(3, 0, 0, math))   # 128 x 128, two workgroups per CU
        if cout >= 192:
            cands.append(_variant(3, 0, 3, math))   # 128 x 256, eight waves
    if math_name == "bf16x3" and 3 in _X3_SCHEDULES and cout >= 192 and cred % 32 == 0 and cred > 0:
        cands.append(_variant(3, 0, 0, math))  # (conv3_covers; a launch it does not cover runs the first schedule)
    return cands


def _wgrad_candidates(cout, cin, math_name=None):
    tms = (0, 1) if cout > 64 else (0,)
    tns = (0, 1) if cin % 128 == 0 else (0,)
    math_name = math_name or _math("wgrad")
    math = _MATH_CODES[math_name]
    bks = (1, 2) if math_name == "f32" else (1,)  # (pixel step; the bf16 / fp16 weight-gradient kernels have one)
    return [(split, _variant(bk, tm, tn, math)) for tm in tms for tn in tns for bk in bks for split in (0, -2, -3, -4)]


# ---- dgrad || wgrad overlap ----------------------------------------------------------------------------------------
# The two gradient kernels of a convolution are independent.  Launched back to back on one stream, each ends with a
# partially filled last round of workgroups (grids are small multiples of the 256 CUs); launched on two streams the
# tails overlap.  Measured on MI355X (bs 8, 320x704): 85.0 ms/step with the overlap vs 83.4 without on the fp32-MFMA
# kernels, 58.2-58.3 vs 57.2-57.6 on the bf16x3 kernels -- two MFMA kernels mostly contend for the same pipes and LDS --
# so it is OFF unless HEATNET_OVERLAP=1.
OVERLAP = os.environ.get("HEATNET_OVERLAP", "0") == "1"
_side_streams = {}


SIDE_PRIORITY = int(os.environ.get("HEATNET_SIDE_PRIORITY", "0"))  # experiment knob: priority of the side stream


def _side_stream(device):
    s = _side_streams.get(device)
    if s is None:
        s = _side_streams[device] = torch.cuda.Stream(device=device, priority=SIDE_PRIORITY)
    return s


_main_streams = {}
# ---- the two encoders of a late-fusion network on two streams ---------------------------------------------------------------
# RGB and thermal encoders (stem, layer1, layer2; cm/models/extractors.py:172-186) are independent up to the concatenation in
# front of layer3: mostly short-reduction 1x1 convolutions and BatchNorm passes over large maps, 30-60 us launches that leave
# the chip half idle at their ramps and tails.  resnet.ResNet.forward runs the thermal chain on a second stream; autograd
# replays every node's backward on its forward stream, so the backward overlaps the same way.  Shared state that had to become
# per-stream: scale-slot chunks, the deterministic weight-gradient workspace; ddp.GradBuckets waits for both streams.
# Measured (profiles/r03_dual_streams.txt, ms per step, one box): second stream 38.95 / N > 1 path on one rank 39.2 / bf16 25.5 /
# train_seg 87.4 against 39.9 / 40.2 / 26.1 / 89.0 with one encoder stream and the weight gradients on a side stream
# (WGRAD_STREAM); both together: 38.8 plain but 50.7 in the N > 1 path -- so the side stream stays off where this one is in use.
DUAL_STREAMS = os.environ.get("HEATNET_DUAL_STREAMS", "1") != "0"
_dual_streams = {}
dual_stream_used = {}  # device -> True once an encoder ran on the second stream (ddp asks)


def dual_stream(device):
    s = _dual_streams.get(device)
    if s is None:
        s = _dual_streams[device] = torch.cuda.Stream(device=device)
    return s


dual_region = [False]  # inside the forward of the two-encoder part of a network (set by its forward around fork and join)
SECOND_STREAM_WGRAD = os.environ.get("HEATNET_SECOND_STREAM_WGRAD", "0") != "0"  # experiment, see _ConvBnAct.backward


def dual_fork(device):
    """(main, second): the second stream waits for everything queued on the current one so far."""
    main = torch.cuda.current_stream(device)
    second = dual_stream(device)
    second.wait_stream(main)
    dual_stream_used[device] = main  # (what ddp.GradBuckets has to wait for besides the hook's own stream)
    dual_region[0] = True
    return main, second


def dual_join(main, second):
    main.wait_stream(second)
    dual_region[0] = False


def _queue_second_stream_join(device, main, second):
    task = _graph_task()
    if _dual_join_task[0] == task:
        return
    _dual_join_task[0] = task

    def join():
        main.wait_stream(second)
        cur = torch.cuda.current_stream(device)
        if cur != main:
            cur.wait_stream(second)
    torch.autograd.Variable._execution_engine.queue_callback(join)


_dual_join_task = [None]


def note_backward_stream(device):
    """Called by the convolution nodes' backward: when this one runs on the second encoder stream, the caller's stream (and the
    main one) wait for that stream once the backward pass ends -- the engine's own end-of-pass synchronisation only follows
    gradients it was handed, not a weight gradient added in place into its arena slot (second use of a layer: returns None)."""
    if not dual_stream_used or device not in dual_stream_used:
        return
    second = _dual_streams.get(device)
    if second is None or torch.cuda.current_stream(device) != second:
        return
    task = _graph_task()
    if _dual_join_task[0] == task:
        return
    _dual_join_task[0] = task
    main = dual_stream_used[device]

    def join():
        main.wait_stream(second)
        cur = torch.cuda.current_stream(device)
        if cur != main:
            cur.wait_stream(second)
    torch.autograd.Variable._execution_engine.queue_callback(join)


def hand_over(t, stream):
    """`t` was produced on another stream and is from now on read on `stream` (which already waits for the producer): tell the
    caching allocator, for the tensor, its plane twin and its scale slot."""
    t.record_stream(stream)
    tw = getattr(t, "_hn_twin", None)
    if tw is not None:
        tw[1].record_stream(stream)
        sl = _plane_slot(getattr(tw[1], "_hn_planes", None)) if is_planes(tw[1]) else None
        if sl is not None:
            sl.record_stream(stream)
    if is_planes(t):
        sl = _plane_slot(t._hn_planes)
        if sl is not None:
            sl.record_stream(stream)
    hit = getattr(t, "_hn_scale", None)
    if hit is not None:
        hit[1].record_stream(stream)



class main_stream:
    """`with ops.main_stream(device): ...` -- experiment knob HEATNET_MAIN_HIGH=1: run the enclosed work on a HIGH-priority
    stream, so that the critical path (BatchNorm passes, data gradients) is dispatched ahead of the weight gradients that
    share the GPU from the side stream.  A no-op without the knob."""
    ON = os.environ.get("HEATNET_MAIN_HIGH", "0") == "1"

    def __init__(self, device):
        self.device = device
        self.ctx = None

    def __enter__(self):
        if not self.ON or self.device.type != "cuda":
            return self
        hp = _main_streams.get(self.device)
        if hp is None:
            hp = _main_streams[self.device] = torch.cuda.Stream(device=self.device, priority=-1)
        self.outer = torch.cuda.current_stream(self.device)
        hp.wait_stream(self.outer)
        self.hp = hp
        self.ctx = torch.cuda.stream(hp)
        self.ctx.__enter__()
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.__exit__(*exc)
            self.outer.wait_stream(self.hp)
        return False


# ---- weight gradients beside the BatchNorm backward passes ---------------------------------------------------------------
# In the backward chain  dz_k -> [BatchNorm_k reduce, apply] -> dy_k -> dgrad_k -> dz_(k-1) -> ...  the weight gradient of layer k
# only needs dy_k and the saved input: it is off the critical path.  The BatchNorm passes are HBM-bound (5+ TB/s, matrix cores
# idle), the weight gradient is bound by the matrix cores / LDS: launched on a second stream, wgrad_k runs WHILE the main stream
# does the BatchNorm backward of layer k-1 -- complementary resources, unlike dgrad || wgrad (HEATNET_OVERLAP above: two matrix
# kernels just contend).  The side stream is joined once, when the backward pass ends (an autograd engine callback), and before
# any gradient bucket is handed to RCCL (ddp.GradBuckets._launch).  HEATNET_WGRAD_STREAM=0 keeps everything on one stream.
WGRAD_STREAM = os.environ.get("HEATNET_WGRAD_STREAM", "1") != "0"
_side_join_task = [None]
side_stream_used = {}  # device -> True once a weight gradient ran on the side stream (ddp asks)


def _wgrad_side_stream(device, main):
    """The side stream for this backward pass's weight gradients, with the join queued at the end of the pass."""
    side = _side_stream(device)
    side_stream_used[device] = True
    task = _graph_task()
    if _side_join_task[0] != task:
        _side_join_task[0] = task

        def join():
            main.wait_stream(side)
            cur = torch.cuda.current_stream(device)
            if cur != main:
                cur.wait_stream(side)
        torch.autograd.Variable._execution_engine.queue_callback(join)
    return side


def join_side_stream(device=None):
    """Make the current stream wait for weight gradients still running on the side stream (explicit form of the join the
    autograd callback does; ddp calls it on its communication stream)."""
    for dev, side in _side_streams.items():
        if device is None or dev == device:
            torch.cuda.current_stream(dev).wait_stream(side)


# ---- gradient slots -------------------------------------------------------------------------------------------------
# optim.ParamArena gives every parameter a fixed slot in a flat gradient buffer that is cleared once per step and marks
# the parameter with `_hn_grad_slot`.  The backward that produces the FIRST gradient of such a parameter in a step
# (param.grad is None) writes it straight into the slot and returns a fresh view of it; autograd's AccumulateGrad then
# adopts that tensor as `.grad` without a copy or an add.  A further contribution in the same step (a layer applied
# twice: day and night passes) is ADDED into the slot by the kernel itself and reported to autograd as "no gradient"
# (None): the leaf's AccumulateGrad -- and any hook on it -- still runs once, after all of its uses.
#
# "First use" is decided per BACKWARD CALL, not from `_hn_slot_used` alone: the flag is reset only by the arena's own
# zero_grad(), and gradients can be dropped in other ways (nn.Module.zero_grad() sets `.grad = None` in torch >= 2,
# `p.grad = None`, an exception in the middle of a backward).  The slot remembers the autograd graph-task id of the
# backward that claimed it: a later use inside the SAME backward adds; a use from ANOTHER backward that finds `.grad`
# gone means the slot holds a dropped gradient -- it is cleared and claimed afresh instead of being added to.
def _graph_task():
    return torch._C._current_graph_task_id()


def grad_slot(param, accumulate_ok=True):
    """(view, first) for `param`'s arena gradient slot -- `first`: the slot is empty (zero) and the caller's result
    becomes the gradient; not first: the caller must add into it -- or None when there is no slot to use."""
    slot = getattr(param, "_hn_grad_slot", None)
    if slot is None:
        return None
    flat, off = slot
    view = flat[off:off + param.numel()].as_strided(param.size(), param.stride())
    task = _graph_task()
    used = getattr(param, "_hn_slot_used", False)
    if param.grad is None and (not used or getattr(param, "_hn_slot_task", None) != task):
        if used:  # gradient dropped behind the arena's back: the slot still holds it
            flat[off:off + param.numel()].zero_()
        param._hn_slot_used = True  # reset by the optimizer's zero_grad()
        param._hn_slot_task = task
        return view, True
    if accumulate_ok and used and (param.grad is None or param.grad.data_ptr() == view.data_ptr()):
        return view, False
    return None


def _leaf(t):
    return t if isinstance(t, torch.nn.Parameter) else None


# Layers with a tiny channel count on one side -- the 7x7 stride-2 stems (1 / 3 / 4 input channels) and the 13-class head --
# run on streaming fp32-MFMA kernels of their own (csrc/conv_small.hip); HEATNET_SMALL_KERNELS=0: the generic kernels (A/B).
SMALL_KERNELS = os.environ.get("HEATNET_SMALL_KERNELS", "1") != "0"
# The stem kernels gather their 7x7 windows straight from global memory (37 scattered 4-byte loads per 16 pixels) and measured
# SLOWER than the generic kernels (forward 0.24 vs 0.14 ms, weight gradient 0.36 vs 0.34 ms, profiles/r03_small_kernels.txt);
# they stay built and tested but are off unless HEATNET_SMALL_STEM=1.  The head kernels are on: weight gradient 0.35 -> 0.15 ms,
# data gradient 0.21 -> 0.12 ms.
SMALL_STEM = os.environ.get("HEATNET_SMALL_STEM", "0") == "1"
# Round 4: the stem WEIGHT GRADIENT fetches dy four pixel groups ahead (a load in front of every group had left the matrix pipe
# idle for a memory latency sixteen times per tile): 0.34 -> 0.28 ms for the RGB stem alone, and with its 70-120 registers per
# lane the kernel shares a CU with the other encoder stream's kernels -- the step got 0.25 ms shorter (gpurun_out/r4/ab7).  On by
# default, independently of the forward kernel (which stays behind HEATNET_SMALL_STEM: 0.23 against 0.14 ms generic).
SMALL_STEM_WGRAD = os.environ.get("HEATNET_SMALL_STEM_WGRAD", "1") != "0"


def _stem_geometry(d):
    return (d.KH == 7 and d.KW == 7 and d.stride == 2 and d.pad == 3 and d.dil == 1 and d.Cout == 64 and d.Cin in (1, 3) and
            d.ldx == d.Cin and d.ldy % 4 == 0)


def _narrow_geometry(d):
    return (d.KH == 1 and d.KW == 1 and d.stride == 1 and d.pad == 0 and d.dil == 1 and d.Cin == 64 and d.Cout <= 16 and
            d.ldx % 4 == 0)


_conv_prof = None  # bench.py: list receiving (kind, algorithmic flops, start event, end event) per conv launch


def set_conv_profiler(records):
    global _conv_prof
    _conv_prof = records


def _plan_name(kind, d, has_planes, variant):
    """The kernel instantiation hn_conv2d_fwd (kind 0) / hn_conv2d_dgrad (kind 1) runs for this geometry, as rocprofv3
    names it (per-instantiation accounting in bench.py)."""
    buf = ctypes.create_string_buffer(128)
    rc = _lib.load().hn_conv2d_plan(kind, ctypes.byref(d), int(has_planes), variant, buf, 128)
    return buf.value.decode() if rc >= 0 else "conv_igemm_kernel<?>"


def _wgrad_plan_name(d, mode, variant):
    """The kernel instantiation hn_conv2d_wgrad (mode 0) / _scaled (1) / _planes (2) runs, as rocprofv3 names it."""
    buf = ctypes.create_string_buffer(128)
    rc = _lib.load().hn_conv2d_wgrad_plan(ctypes.byref(d), mode, variant, buf, 128)
    return buf.value.decode() if rc >= 0 else "conv_wgrad_kernel<?>"


class _ConvTimer:
    """HIP events on the launch stream (torch's current stream) around one conv-kernel launch."""

    def __init__(self, kind, d, kname=None, extra_bytes=0.0):
        self.on = _conv_prof is not None
        if self.on:
            self.kind = kind
            self.kname = kname() if callable(kname) else (kname or kind)
            self.flops = 2.0 * d.N * d.Hout * d.Wout * d.Cout * d.KH * d.KW * d.Cin
            self.shape = (d.N, d.Hin, d.Win, d.Cin, d.Hout, d.Wout, d.Cout, d.KH, d.stride, d.dil)
            # compulsory HBM traffic: each of the three tensors of the convolution touched once
            # (+ extra_bytes: what the launch's fused epilogue reads on top -- the gradient of the by-passing branch it adds,
            # the BatchNorm output whose backward reduction rides along)
            self.bytes = 4.0 * (d.N * d.Hin * d.Win * d.Cin + d.Cout * d.KH * d.KW * d.Cin + d.N * d.Hout * d.Wout * d.Cout) + extra_bytes
            self.e0, self.e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def __enter__(self):
        if self.on:
            self.e0.record()
        return self

    def __exit__(self, *exc):
        if self.on:
            self.e1.record()
            _conv_prof.append((self.kind, self.flops, self.e0, self.e1, self.shape, self.bytes, self.kname))


def _weight_operands(w_in, w, stride, pad, dil, lm, need_dgrad, s):
    """(planes of w, flipped / transposed fp32 weights, their planes) for a layer whose arithmetic is `lm` -- (None, None, None)
    under "f32".  Weight operands of the bf16x3 / bf16 / f16x2 kernels: the 16-bit planes of w for the forward and -- when a data
    gradient will follow -- the flipped / transposed weights (fp32 and planes) for dgrad, all made by ONE launch.  They are
    kept on the parameter and reused inside a frozen_weights() scope (second pass of a confusion-maximisation step,
    evaluation loops)."""
    if lm not in _PLANE_MATHS:
        return None, None, None
    lib = _lib.load()
    cout, cin, kh, kw = w.shape
    fmt = 1 if lm == "f16x2" else 0
    want_t = _layer_math("dgrad", cin, cout, kh, kw, _leaf(w_in)) in _PLANE_MATHS and need_dgrad
    leaf = _leaf(w_in) if w is w_in else None
    key = (w.data_ptr(), w._version, _frozen_scope, stride, pad, dil, fmt)
    prep = getattr(leaf, "_hn_prep", None) if (leaf is not None and _frozen_depth > 0) else None
    if prep is None or prep[0] != key or (want_t and prep[2] is None):
        nw = w.numel()
        wx3 = torch.empty(3 * x3_plane(nw), device=w.device, dtype=torch.int16)
        if want_t:
            wt = torch.empty(nw, device=w.device, dtype=torch.float32)
            wtx3 = torch.empty(3 * x3_plane(nw), device=w.device, dtype=torch.int16)
            flip = lib.hn_weight_flip_transpose_h2 if fmt else lib.hn_weight_flip_transpose
            check(flip(_ptr(w), _ptr(wx3), _ptr(wt), _ptr(wtx3), cout, kh, kw, cin, stride, pad, dil, s),
                  "hn_weight_flip_transpose")
        else:
            wt = wtx3 = None
            check((lib.hn_split_f16x2 if fmt else lib.hn_split_bf16x3)(_ptr(w), _ptr(wx3), nw, s), "hn_split_planes")
        prep = (key, wx3, wt, wtx3)
        if leaf is not None and _frozen_depth > 0:
            leaf._hn_prep = prep
            _prep_registry.touch(leaf, (stride, pad, dil), wx3, wt, wtx3, fmt)  # next scope: one batched launch
    elif leaf is not None:
        _prep_registry.touch(leaf, (stride, pad, dil), prep[1], prep[2], prep[3], fmt)
    return prep[1], prep[2], prep[3]


# ------------------------------------------------------------------------------------------------------
# convolution
# ------------------------------------------------------------------------------------------------------

class _Conv2d(torch.autograd.Function):
    """y = act(conv(x, w) + b) [+ per-channel batch statistics of the pre-activation output]."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dil, act, slope, want_stats, passthrough, zero_bias_grad=False):
        lib = _lib.load()
        x_in, w_in = x, w
        x, ldx = as_nhwc(x)
        w = weight_ohwi(w)
        n, cin, h, wd = x.shape
        cout, cin_w, kh, kw = w.shape
        if cin_w != cin:
            raise RuntimeError("conv2d: input has %d channels, weight expects %d" % (cin, cin_w))
        ho, wo = conv_out_size(h, kh, stride, pad, dil), conv_out_size(wd, kw, stride, pad, dil)
        y = new_nhwc(n, cout, ho, wo, x.device)
        if want_stats is True:
            stats = torch.zeros(STAT_REPLICAS * 2 * cout, device=x.device, dtype=torch.float64)
        elif torch.is_tensor(want_stats):
            stats = want_stats  # persistent per-layer accumulator: zero on entry, re-zeroed by hn_bn_finalize
        else:
            stats = None
        d = _desc(n, h, wd, cin, ldx, ho, wo, cout, cout, kh, kw, stride, pad, dil)
        s = _stream()
        lm = _layer_math("fwd", cin, cout, kh, kw, _leaf(w_in))  # this layer's arithmetic (one plane format for all kinds)
        wx3, ctx.wt, ctx.wtx3 = _weight_operands(w_in, w, stride, pad, dil, lm, ctx.needs_input_grad[0], s)
        ctx.lm = lm
        ctx.x_scale = None
        ctx.small = None
        if SMALL_KERNELS and SMALL_STEM_WGRAD and _stem_geometry(d):
            ctx.small = "stem"  # (the weight gradient; the forward below only with HEATNET_SMALL_STEM)
        stem_fwd = SMALL_KERNELS and SMALL_STEM and _stem_geometry(d) and b is None and act == ACT_NONE and want_stats is not True
        if stem_fwd:
            # 7x7 / stride-2 stem with 1, 3 or 4 input channels: its own fp32-MFMA streaming kernel (csrc/conv_small.hip)
            ctx.small = "stem"
            with _ConvTimer("fwd", d, "stem_fwd_kernel<%d>" % cin):
                check(lib.hn_conv_stem_fwd(_ptr(x), _ptr(w), _ptr(y), _ptr(stats), ctypes.byref(d), s), "hn_conv_stem_fwd")
        elif lm == "f16x2" and _h2_covers(0, d) and x.data_ptr() % 16 == 0:
            xs = ctx.x_scale = tensor_scale(x, ldx, owner=x_in)
            if RANGE_CENSUS:
                _census_operand(_leaf(w_in), 0, x, None, ldx, xs, n * h * wd, cin)

            def launch(v, st):
                check(lib.hn_conv2d_fwd_scaled(_ptr(x), _ptr(xs), _ptr(w), _ptr(wx3), _ptr(b), _ptr(y), _ptr(st),
                                               ctypes.byref(d), act, slope, v, s), "hn_conv2d_fwd_scaled")
        else:
            if lm == "f16x2":
                lm, wx3 = "f32", None  # a geometry the f16x2 kernels do not cover (the fp16 planes are of no use to the others)

            def launch(v, st):
                check(lib.hn_conv2d_fwd(_ptr(x), _ptr(w), _ptr(wx3), _ptr(b), _ptr(y), _ptr(st), ctypes.byref(d), act, slope,
                                        v, s), "hn_conv2d_fwd")
        if ctx.small is None or (ctx.small == "stem" and not stem_fwd):
            if SMALL_KERNELS and _narrow_geometry(d):
                ctx.small = "narrow"  # (13-class head: forward on the generic kernel -- already at its HBM time --, backward below)
            var = _pick(_kind("fwd", lm), d, _igemm_candidates(cout, "fwd", cin, lm), lambda v: launch(v, None))
            with _ConvTimer("fwd", d, lambda: _plan_name(0, d, wx3 is not None, var)):
                launch(var, stats)
        ctx.geom = (stride, pad, dil, act, slope)
        ctx.has_bias = b is not None
        ctx.w_leaf, ctx.b_leaf = _leaf(w_in), _leaf(b)
        ctx.zero_bias_grad = zero_bias_grad
        ctx.save_for_backward(x, w, y if act != ACT_NONE else None)
        ctx.passthrough = passthrough
        if passthrough:
            # second output = the input itself (an alias): whatever consumes it (the block's skip connection) sends
            # its gradient back through THIS node, where it is added in the dgrad epilogue
            return y, x_in.view_as(x_in)
        if want_stats is True:
            ctx.mark_non_differentiable(stats)
            return y, stats
        return y  # a caller-owned accumulator is not returned (it is an input)

    @staticmethod
    def backward(ctx, dy, *extra):
        lib = _lib.load()
        x, w, y = ctx.saved_tensors
        note_backward_stream(x.device)
        dskip = extra[0] if (ctx.passthrough and extra) else None  # gradient that by-passed this conv
        stride, pad, dil, act, slope = ctx.geom
        x, ldx = as_nhwc(x)
        dy, lddy = as_nhwc(dy)
        n, cin, h, wd = x.shape
        cout, _, kh, kw = w.shape
        _, _, ho, wo = dy.shape
        npix = n * ho * wo
        s = _stream()
        ctx.db_done = None
        if act != ACT_NONE:
            dpre = new_nhwc(n, cout, ho, wo, x.device)
            if ctx.has_bias and ctx.needs_input_grad[2] and not ctx.zero_bias_grad:
                # the bias gradient is the column sum of the tensor this pass writes: taken along (one pass over dy, not two)
                slot = grad_slot(ctx.b_leaf, accumulate_ok=False) if ctx.b_leaf is not None else None
                db = slot[0] if slot is not None else torch.empty(cout, device=x.device, dtype=torch.float32)
                dslot = _new_scale_slot(x.device) if ctx.lm == "f16x2" else None  # (the scale the backward kernels split dpre with)
                check(lib.hn_act_bwd_bias(_ptr(dy), lddy, _ptr(y), cout, _ptr(dpre), cout, npix, cout, act, slope, _ptr(db),
                                          _ptr(dslot), s), "hn_act_bwd_bias")
                if dslot is not None:
                    dpre._hn_scale = (dpre._version, dslot)
                ctx.db_done = db
            else:
                check(lib.hn_act_bwd(_ptr(dy), lddy, _ptr(y), cout, _ptr(dpre), cout, npix, cout, act, slope, s),
                      "hn_act_bwd")
            dy, lddy = dpre, cout
        d = _desc(n, h, wd, cin, ldx, ho, wo, cout, lddy, kh, kw, stride, pad, dil)
        dx = dw = db = None
        fork = None
        lm = ctx.lm  # the layer's arithmetic as decided (and its weight planes made) at forward time
        h2_layer = lm == "f16x2"
        dy_scale = None
        if OVERLAP and _conv_prof is None and ctx.needs_input_grad[0] and ctx.needs_input_grad[1]:
            if h2_layer:
                # everything the side-stream weight gradient reads must be queued BEFORE the fork event: the operand scales
                # (an hn_absmax_scale pass, and the zero fill of a fresh slot chunk, run on the main stream)
                if dy.data_ptr() % 16 == 0:
                    dy_scale = tensor_scale(dy, lddy)
                if ctx.x_scale is None and x.data_ptr() % 16 == 0:
                    ctx.x_scale = tensor_scale(x, ldx)
            fork = torch.cuda.Event()
            fork.record()  # dy (and the activation backward that produced it) and the scales are complete up to here
        small = ctx.small if not DETERMINISTIC else None  # (the streaming kernels reduce with float atomics)
        if ctx.needs_input_grad[0] and small == "narrow" and dskip is None:
            dx = new_nhwc(n, cin, h, wd, x.device)
            dn = _desc(n, h, wd, cin, cin, ho, wo, cout, lddy, kh, kw, stride, pad, dil)
            with _ConvTimer("dgrad", dn, "narrow_dgrad_kernel"):
                check(lib.hn_conv_narrow_dgrad(_ptr(dy), _ptr(w), _ptr(dx), ctypes.byref(dn), s), "hn_conv_narrow_dgrad")
        elif ctx.needs_input_grad[0]:
            nw = cin * kh * kw * cout
            wt, wtx3 = ctx.wt, ctx.wtx3  # prepared by the forward (plane arithmetics), else made here
            dm = lm if lm != "f16x2" else _layer_math("dgrad", cin, cout, kh, kw, ctx.w_leaf)
            if wt is None:
                wt = torch.empty(nw, device=x.device, dtype=torch.float32)
                wtx3 = torch.empty(3 * x3_plane(nw), device=x.device, dtype=torch.int16) if dm in _PLANE_MATHS else None
                flip = lib.hn_weight_flip_transpose_h2 if (h2_layer and wtx3 is not None) else lib.hn_weight_flip_transpose
                check(flip(_ptr(w), None, _ptr(wt), _ptr(wtx3), cout, kh, kw, cin, stride, pad, dil, s),
                      "hn_weight_flip_transpose")
            dx = new_nhwc(n, cin, h, wd, x.device)
            dd = _desc(n, h, wd, cin, cin, ho, wo, cout, lddy, kh, kw, stride, pad, dil)
            add, ldadd = (None, 0) if dskip is None else as_nhwc(dskip)
            if h2_layer and wtx3 is not None and _h2_covers(1, dd) and dy.data_ptr() % 16 == 0:
                dy_scale = dy_scale if dy_scale is not None else tensor_scale(dy, lddy)
                if RANGE_CENSUS:
                    _census_operand(ctx.w_leaf, 1, dy, None, lddy, dy_scale, npix, cout)

                def launch(v, ad, ld):
                    check(lib.hn_conv2d_dgrad_scaled(_ptr(dy), _ptr(dy_scale), _ptr(wt), _ptr(wtx3), _ptr(dx), _ptr(ad), ld,
                                                     ctypes.byref(dd), v, s), "hn_conv2d_dgrad_scaled")
            else:
                if h2_layer:
                    dm, wtx3 = "f32", None  # not covered: the fp32 MFMA on the fp32 copy of the operand

                def launch(v, ad, ld):
                    check(lib.hn_conv2d_dgrad(_ptr(dy), _ptr(wt), _ptr(wtx3), _ptr(dx), _ptr(ad), ld, ctypes.byref(dd), v, s),
                          "hn_conv2d_dgrad")
            var = _pick(_kind("dgrad", dm), dd, _igemm_candidates(cin, "dgrad", cout, dm), lambda v: launch(v, None, 0))
            with _ConvTimer("dgrad", dd, lambda: _plan_name(1, dd, wtx3 is not None, var),
                            4.0 * n * h * wd * cin if add is not None else 0.0):
                launch(var, add, ldadd)
        elif dskip is not None:
            dx = dskip
        dw_out = None
        if ctx.needs_input_grad[1]:
            slot = grad_slot(ctx.w_leaf) if ctx.w_leaf is not None else None
            flags = 0
            if slot is not None and weight_ohwi(slot[0]) is slot[0]:
                dw = slot[0]
                flags = 1 if slot[1] else 2  # HN_WGRAD_DW_ZEROED (arena cleared by zero_grad) | HN_WGRAD_ACCUMULATE
                dw_out = dw if slot[1] else None  # a later use adds in place and reports no gradient of its own
            else:
                dw_out = dw = torch.empty((cout, kh, kw, cin), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
            if (small == "stem" and dy.data_ptr() % 16 == 0 and lddy % 4 == 0) or (small == "narrow" and x.data_ptr() % 16 == 0):
                fn = lib.hn_conv_stem_wgrad if small == "stem" else lib.hn_conv_narrow_wgrad
                name = ("stem_wgrad_kernel<%d>" % cin) if small == "stem" else "narrow_wgrad_kernel"
                with _ConvTimer("wgrad", d, name):
                    check(fn(_ptr(x), _ptr(dy), _ptr(dw), ctypes.byref(d), flags, s), "hn_conv_small_wgrad")
                return _Conv2d._finish_backward(ctx, lib, dx, dw_out, dy, lddy, npix, cout, x, s)
            # f16x2 layer: the weight gradient too where its kernel covers the geometry (whole 64-channel blocks both ways),
            # with the input's scale from the forward and dY's from the data gradient above; elsewhere as under bf16x3
            wm = lm if not h2_layer else _layer_math("wgrad", cin, cout, kh, kw, ctx.w_leaf)
            h2_w = (h2_layer and wm == "f16x2" and cin % 64 == 0 and cout % 64 == 0 and ldx % 4 == 0 and lddy % 4 == 0 and
                    x.data_ptr() % 16 == 0 and dy.data_ptr() % 16 == 0 and
                    ((n * h * wd - 1) * ldx + cin) * 4 < 0xFFFF0000 and ((npix - 1) * lddy + cout + 64 * lddy) * 4 < 0xFFFF0000)
            if h2_w:
                x_scale = ctx.x_scale if ctx.x_scale is not None else tensor_scale(x, ldx)
                dy_scale = dy_scale if dy_scale is not None else tensor_scale(dy, lddy)

                def wgrad(dst, sp, v, fl, stream=s):
                    check(lib.hn_conv2d_wgrad_scaled(_ptr(x), _ptr(x_scale), _ptr(dy), _ptr(dy_scale), _ptr(dst), ctypes.byref(d),
                                                     sp, v, fl, stream), "hn_conv2d_wgrad_scaled")

                def wgrad_det(sp, v, fl, wsp):
                    check(lib.hn_conv2d_wgrad_det_scaled(_ptr(x), _ptr(x_scale), _ptr(dy), _ptr(dy_scale), _ptr(dw),
                                                         ctypes.byref(d), sp, v, fl, _ptr(wsp), s), "hn_conv2d_wgrad_det_scaled")
            else:
                if h2_layer:
                    wm = "bf16x3"  # (the library itself falls back to the fp32 MFMA where bf16x3 does not cover either)

                def wgrad(dst, sp, v, fl, stream=s):
                    check(lib.hn_conv2d_wgrad(_ptr(x), _ptr(dy), _ptr(dst), ctypes.byref(d), sp, v, fl, stream), "hn_conv2d_wgrad")

                def wgrad_det(sp, v, fl, wsp):
                    check(lib.hn_conv2d_wgrad_det(_ptr(x), _ptr(dy), _ptr(dw), ctypes.byref(d), sp, v, fl, _ptr(wsp), s),
                          "hn_conv2d_wgrad_det")
            if DETERMINISTIC:
                # fixed plan: one reduction split per 4096 output pixels (at most 32), tile choice left to the library
                splits = max(1, min(32, npix // 4096))
                wsp = _det_workspace(x.device, splits * dw.numel())
                var = _variant(1, 0, 0, _MATH_CODES[wm])
                with _ConvTimer("wgrad", d, lambda: _wgrad_plan_name(d, 1 if h2_w else 0, var)):
                    wgrad_det(splits, var, flags & 2, wsp)
                split = None
            else:
                split, var = _pick(_kind("wgrad", wm), d, _wgrad_candidates(cout, cin, wm),
                                   lambda sv: wgrad(_tune_scratch(dw) if flags else dw, sv[0], sv[1], 0))
            if split is None:
                pass
            elif OVERLAP and _conv_prof is None and ctx.needs_input_grad[0]:
                # the dgrad above is already queued on the main stream; wgrad runs beside it on the side stream
                main, side = torch.cuda.current_stream(), _side_stream(x.device)
                side.wait_event(fork)
                wgrad(dw, split, var, flags, ctypes.c_void_p(side.cuda_stream))
                main.wait_stream(side)
            else:
                with _ConvTimer("wgrad", d, lambda: _wgrad_plan_name(d, 1 if h2_w else 0, var)):
                    wgrad(dw, split, var, flags)
        return _Conv2d._finish_backward(ctx, lib, dx, dw_out, dy, lddy, npix, cout, x, s)

    @staticmethod
    def _finish_backward(ctx, lib, dx, dw_out, dy, lddy, npix, cout, x, s):
        """Bias gradient and the return tuple of backward()."""
        db = None
        if getattr(ctx, "db_done", None) is not None:
            return dx, dw_out, ctx.db_done, None, None, None, None, None, None, None, None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            slot = grad_slot(ctx.b_leaf, accumulate_ok=False) if ctx.b_leaf is not None else None
            db = slot[0] if slot is not None else None
            if ctx.zero_bias_grad:
                # the caller normalises this output with batch statistics: the bias cancels, its gradient is exactly 0
                db = db if db is not None else torch.zeros(cout, device=x.device, dtype=torch.float32)
                return dx, dw_out, db, None, None, None, None, None, None, None, None
            if db is None:
                db = torch.empty(cout, device=x.device, dtype=torch.float32)
            check(lib.hn_channel_sum(_ptr(dy), lddy, npix, cout, _ptr(db), s), "hn_channel_sum")
        return dx, dw_out, db, None, None, None, None, None, None, None, None


PAD_CIN = os.environ.get("HEATNET_PAD_CIN", "1") != "0"  # A/B knob, see conv2d


class _PadW(torch.autograd.Function):
    """w [Cout][KH][KW][Cin] -> [cout_p][KH][KW][cin_p] with zero output / input channels appended (the corner block comes back
    as gradient)."""

    @staticmethod
    def forward(ctx, w, cout_p, cin_p):
        w = weight_ohwi(w)
        cout, cin, kh, kw = w.shape
        wp = torch.zeros((cout_p, kh, kw, cin_p), device=w.device, dtype=torch.float32).permute(0, 3, 1, 2)
        wp[:cout, :cin].copy_(w)
        ctx.shape = (cout, cin)
        return wp

    @staticmethod
    def backward(ctx, g):
        cout, cin = ctx.shape
        return g[:cout, :cin].contiguous(memory_format=torch.channels_last), None, None


class _PadVec(torch.autograd.Function):
    @staticmethod
    def forward(ctx, b, n_p):
        out = torch.zeros(n_p, device=b.device, dtype=torch.float32)
        out[:b.numel()].copy_(b)
        ctx.n = b.numel()
        return out

    @staticmethod
    def backward(ctx, g):
        return g[:ctx.n].contiguous(), None


class _NarrowChannels(torch.autograd.Function):
    """y_p[:, :c] of an NHWC tensor (no copy); the gradient comes back zero-padded to the full width (one strided copy)."""

    @staticmethod
    def forward(ctx, y_p, c):
        ctx.cp = y_p.shape[1]
        return y_p.detach()[:, :c]

    @staticmethod
    def backward(ctx, g):
        g, ldg = as_nhwc(g)
        n, c, h, w = g.shape
        gp = torch.zeros((n, h, w, ctx.cp), device=g.device, dtype=torch.float32).permute(0, 3, 1, 2)
        check(_lib.load().hn_copy(_ptr(g), ldg, _ptr(gp), ctx.cp, n * h * w, c, _stream()), "hn_copy")
        return gp, None


def conv2d(x, w, b=None, stride=1, pad=0, dil=1, act=ACT_NONE, slope=0.0, want_stats=False, passthrough=False,
           zero_bias_grad=False):
    """`want_stats`: False, True (fresh accumulator) or a zeroed float64 [STAT_REPLICAS*2*Cout] accumulator.
    `passthrough=True` returns (y, x) with x an alias of the input whose gradient is folded into this conv's dgrad
    (needs a caller-owned accumulator or no statistics).  `zero_bias_grad=True`: the caller feeds y into a BatchNorm that
    uses batch statistics, where a per-channel bias cancels exactly -- the bias gradient is returned as exact zeros
    instead of the rounding noise a full reduction pass over dy would produce."""
    if passthrough and want_stats is True:
        raise ValueError("conv2d: passthrough cannot be combined with want_stats=True")
    cout, cin = w.shape[0], w.shape[1]
    pad_in = cin >= 128 and cin % 32 != 0
    pad_out = cout >= 128 and cout % 32 != 0  # (the gathered operand of the data / weight gradient is dY)
    if PAD_CIN and x.is_cuda and (pad_in or pad_out) and not passthrough and x.dim() == 4 and x.shape[1] == cin:
        # A wide layer whose channel count is not a multiple of 32 (the 525-channel decoder of the ResNeXt branch,
        # cm/models/trgb_segnet.py:201-209) would run on the scalar-gather fp32 kernels at a sixth of the rate: zero channels
        # are appended to the input, to the weights (both ways) and to the bias -- one copy of each -- the convolution runs at
        # the padded widths and the output is the narrow view of its result; the same sums.
        # (to whole 64-channel blocks: the weight-gradient kernels' vector paths want them on both sides)
        cin_p, cout_p = (cin + 63) // 64 * 64 if pad_in else cin, (cout + 63) // 64 * 64 if pad_out else cout
        if cin_p != cin:
            n_, _, h_, w_ = x.shape
            zeros = torch.zeros((n_, h_, w_, cin_p - cin), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
            x = concat([x, zeros])
        w = _PadW.apply(w, cout_p, cin_p)
        if cout_p == cout:
            return conv2d(x, w, b, stride, pad, dil, act, slope, want_stats, False, zero_bias_grad)
        bp = _PadVec.apply(b, cout_p) if b is not None else None
        res = _Conv2d.apply(x, w, bp, int(stride), int(pad), int(dil), int(act), float(slope), bool(want_stats is not False), False,
                            bool(zero_bias_grad))
        y_p, st_p = res if want_stats is not False else (res, None)
        y = _NarrowChannels.apply(y_p, cout)
        if want_stats is False:
            return y
        st = st_p.view(STAT_REPLICAS, 2, cout_p)[:, :, :cout]
        if want_stats is True:
            return y, st.contiguous().view(-1)
        want_stats.view(STAT_REPLICAS, 2, cout).add_(st)  # the caller's accumulator (zero on entry)
        return y
    out = _Conv2d.apply(x, w, b, int(stride), int(pad), int(dil), int(act), float(slope), want_stats, bool(passthrough),
                        bool(zero_bias_grad))
    if passthrough:
        _inherit_scale(out[1], x)  # the alias of the input carries the input's f16x2 scale (found by this call at the latest)
    return out


# ------------------------------------------------------------------------------------------------------
# batch norm (+ residual) (+ activation)
# ------------------------------------------------------------------------------------------------------

RELU_BITS = os.environ.get("HEATNET_RELU_BITS", "1") != "0"  # A/B knob: byte ReLU masks for the residual BatchNorm layers


class _BnAct(torch.autograd.Function):
    """z = act(bn(y) + residual).  Training mode takes the batch statistics produced by the conv epilogue
    (or computes them when `stats` is None) and updates the running buffers in place, as nn.BatchNorm2d does."""

    @staticmethod
    def forward(ctx, y, stats, gamma, beta, running_mean, running_var, residual, prelu_w, training, momentum, eps,
                act, leaky, ws, drop_mask, z_scale=None):
        lib = _lib.load()
        y, ldy = as_nhwc(y)
        n, c, h, w = y.shape
        npix = n * h * w
        s = _stream()
        dev = y.device
        par = torch.empty(4 * c, device=dev, dtype=torch.float32)  # mean | invstd | scale | shift
        mean, invstd, scale, shift = par[0:c], par[c:2 * c], par[2 * c:3 * c], par[3 * c:4 * c]
        if training:
            if stats is None:
                if ws is not None:
                    ws.open_stats()
                stats = ws.stats if ws is not None else torch.zeros(STAT_REPLICAS * 2 * c, device=dev,
                                                                    dtype=torch.float64)
                check(lib.hn_channel_stats(_ptr(y), ldy, npix, c, _ptr(stats), s), "hn_channel_stats")
            check(lib.hn_bn_finalize(_ptr(stats), npix, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                     momentum, eps, _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), c, s),
                  "hn_bn_finalize")
            if ws is not None:
                ws.stats_open = False  # consumed and re-zeroed
        else:
            check(lib.hn_bn_eval_params(_ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var), eps,
                                        _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), c, s), "hn_bn_eval_params")
        ldr = 0
        if residual is not None:
            if act in (ACT_PRELU, ACT_LEAKY):
                raise RuntimeError("bn_act: a residual input is only supported with ReLU / no activation")
            residual, ldr = as_nhwc(residual)
        z = new_nhwc(n, c, h, w, dev)
        if drop_mask is not None:
            drop_mask = drop_mask.contiguous()
        # ReLU behind a residual add: the backward passes need the sign of z only -- one byte per four channels instead of z
        rbits = None
        if (act == ACT_RELU and residual is not None and c % 4 == 0 and ldy % 4 == 0 and ldr % 4 == 0 and RELU_BITS and
                any(ctx.needs_input_grad) and y.data_ptr() % 16 == 0 and residual.data_ptr() % 16 == 0):
            rbits = torch.empty(npix * (c // 4), device=dev, dtype=torch.uint8)
        check(lib.hn_bn_act_fwd(_ptr(y), ldy, _ptr(scale), _ptr(shift), _ptr(residual), ldr, _ptr(z), c, npix, c, act,
                                _ptr(prelu_w), leaky, _ptr(drop_mask), h * w, _ptr(z_scale), _ptr(rbits), s), "hn_bn_act_fwd")
        ctx.drop_mask = drop_mask
        ctx.leaves = (_leaf(gamma), _leaf(beta), _leaf(prelu_w))
        ctx.cfg = (training, act, leaky, residual is not None)
        ctx.ws = ws
        # the ReLU mask is recomputed from y in backward unless a residual entered u (then z is kept)
        ctx.save_for_backward(y, z if (act == ACT_RELU and residual is not None and rbits is None) else None, par, gamma,
                              prelu_w, rbits)
        return z

    @staticmethod
    def backward(ctx, dz):
        lib = _lib.load()
        y, z, par, gamma, prelu_w, rbits = ctx.saved_tensors
        training, act, leaky, has_res = ctx.cfg
        y, ldy = as_nhwc(y)
        dz, lddz = as_nhwc(dz)
        if rbits is not None:
            dz, lddz = _vector_grad(dz, lddz)
        n, c, h, w = y.shape
        npix = n * h * w
        s = _stream()
        mean, invstd, scale, shift = par[0:c], par[c:2 * c], par[2 * c:3 * c], par[3 * c:4 * c]
        ldz = c
        ws = ctx.ws
        if ws is not None:
            ws.open_bwd()
        sums = ws.bwd if ws is not None else torch.zeros(STAT_REPLICAS * (2 * c + 1), device=y.device,
                                                         dtype=torch.float64)
        dpar = torch.empty(2 * c + 1, device=y.device, dtype=torch.float32)   # dbeta | dgamma | dslope
        # first gradient of a parameter in this step: written straight into its arena slot (see grad_slot)
        lg, lb, lp = ctx.leaves
        sg = grad_slot(lg) if (lg is not None and ctx.needs_input_grad[2]) else None
        sb = grad_slot(lb) if (lb is not None and ctx.needs_input_grad[3]) else None
        sp = grad_slot(lp) if (lp is not None and prelu_w is not None and ctx.needs_input_grad[7]) else None
        used = [t for t in (sg, sb, sp) if t is not None]
        if len({first for _, first in used}) > 1:
            sg = sb = sp = None  # the layer's parameters disagree on first / later use (not expected): plain outputs
            used = []
        accumulate = bool(used) and not used[0][1]  # a later use of the layer in this step: add into the slots
        check(lib.hn_bn_act_bwd_reduce(_ptr(dz), lddz, _ptr(z), ldz, _ptr(y), ldy, _ptr(mean), _ptr(invstd), _ptr(scale),
                                       _ptr(shift), npix, c, act, _ptr(prelu_w), leaky, _ptr(sums), _ptr(dpar),
                                       _ptr(sg[0]) if sg else None, _ptr(sb[0]) if sb else None,
                                       _ptr(sp[0]) if sp else None, int(accumulate), _ptr(ctx.drop_mask), h * w, _ptr(rbits), s),
              "hn_bn_act_bwd_reduce")
        if ws is not None:
            ws.bwd_open = False  # hn_bn_act_bwd_reduce consumed and re-zeroed its accumulator
        dy = new_nhwc(n, c, h, w, y.device) if ctx.needs_input_grad[0] else None
        dres = new_nhwc(n, c, h, w, y.device) if (has_res and ctx.needs_input_grad[6]) else None
        if dy is not None or dres is not None:
            if dy is None:
                dy = new_nhwc(n, c, h, w, y.device)
            # f16x2: the pass also publishes dy's scale (a pass of its own costs as much as this one's read of dz)
            dy_scale = _new_scale_slot(y.device) if (CONV_MATH == "f16x2" and ctx.needs_input_grad[0]) else None
            check(lib.hn_bn_act_bwd_apply(_ptr(dz), lddz, _ptr(z), ldz, _ptr(y), ldy, _ptr(mean), _ptr(invstd),
                                          _ptr(gamma), _ptr(scale), _ptr(shift), _ptr(dpar), npix, int(training),
                                          _ptr(dy), c, _ptr(dres), c, npix, c, act, _ptr(prelu_w), leaky,
                                          _ptr(ctx.drop_mask), h * w, _ptr(dy_scale), _ptr(rbits), s), "hn_bn_act_bwd_apply")
            if dy_scale is not None:
                dy._hn_scale = (dy._version, dy_scale)  # rides with the tensor to the producing convolution's backward
        def out(slot, fallback, needed):  # first use: the slot view; later use: nothing (added in place); no slot: dpar
            if not needed:
                return None
            if slot is None:
                return fallback
            return slot[0] if slot[1] else None

        dgamma = out(sg, dpar[c:2 * c], ctx.needs_input_grad[2])
        dbeta = out(sb, dpar[0:c], ctx.needs_input_grad[3])
        dprelu = out(sp, dpar[2 * c:2 * c + 1], prelu_w is not None and ctx.needs_input_grad[7])
        return dy, None, dgamma, dbeta, None, None, dres, dprelu, None, None, None, None, None, None, None, None


class BnWorkspace:
    """Persistent per-layer double accumulators (zero between uses: the kernels that read them re-zero them)."""

    def __init__(self, c, device):
        self.device = device
        buf = torch.zeros(STAT_REPLICAS * (4 * c + 1), device=device, dtype=torch.float64)
        self.stats, self.bwd = buf[:STAT_REPLICAS * 2 * c], buf[STAT_REPLICAS * 2 * c:]
        # plane-tensor path: range keys of the conv output (forward) and per-channel max |du| (backward); zero between uses
        # (consumed and re-zeroed by hn_bn_finalize_range / hn_bn_act_bwd_reduce_range; a stale value only loosens a bound)
        rng = torch.zeros(STAT_REPLICAS * 3 * c, device=device, dtype=torch.int32)
        self.keys, self.gmax = rng[:STAT_REPLICAS * 2 * c], rng[STAT_REPLICAS * 2 * c:]
        # host-side guards: set while an accumulator may hold partial sums (between the kernel that adds into it and the
        # kernel that consumes + re-zeroes it); if an exception interrupts that window, the next user clears it first
        self.stats_open = self.bwd_open = False
        # the _BnLink whose consumer filled `bwd` / `gmax` from its data-gradient epilogue (between that launch and the layer's
        # own backward); a forward never sees one unless a backward was interrupted
        self.fused_owner = None
        self.fuse_refused = False  # set once a fused reduction had to be discarded (the gradient was a sum of several)

    def open_stats(self):
        if self.stats_open:
            self.stats.zero_()
        if self.fused_owner is not None:
            self.fused_owner = None
            self.bwd.zero_()
            self.gmax.zero_()
            self.bwd_open = False
        self.stats_open = True

    def open_bwd(self):
        if self.bwd_open:
            self.bwd.zero_()
        self.bwd_open = True


def bn_act(y, stats, gamma, beta, running_mean, running_var, residual=None, prelu_weight=None, training=True,
           momentum=0.1, eps=1e-5, act=ACT_NONE, leaky=0.0, ws=None, drop_mask=None):
    """`drop_mask` [N, C]: Dropout2d keep/(1-p) factors applied to the output in the same pass (see `drop_mask()`)."""
    # f16x2: the pass also publishes the output's scale, remembered on the tensor for the convolution that consumes it
    z_scale = _new_scale_slot(y.device) if (CONV_MATH == "f16x2" and y.is_cuda) else None
    z = _BnAct.apply(y, stats, gamma, beta, running_mean, running_var, residual, prelu_weight, bool(training),
                     float(momentum), float(eps), int(act), float(leaky), ws, drop_mask, z_scale)
    if z_scale is not None:
        z._hn_scale = (z._version, z_scale)
    return z



# ------------------------------------------------------------------------------------------------------
# plane tensors + the fused conv -> BatchNorm -> activation node (f16x2 arithmetic)
# ------------------------------------------------------------------------------------------------------
# A PLANE TENSOR is a torch.Tensor with the logical shape, dtype (fp32) and NHWC strides of the activation it stands for whose
# BYTES are two fp16 images [2][N*H*W][C]: hi = fp16(t * s), lo = fp16(t * s - hi), s the power of two derived from the scale
# slot in `t._hn_planes` (include/heatnet_hip.h, "PLANE TENSORS").  It is what the f16x2 kernels otherwise build in registers
# for every tile that touches an element; written once by the pass that produces the tensor, the convolutions that consume it
# (csrc/conv_igemm4.hip, conv_wgrad4 in csrc/conv_wgrad.hip) are pure LDS-DMA + MFMA loops.  Only `conv_bn_act` and the
# residual input of a fused BatchNorm accept plane tensors; every other operator refuses them (`_require_gpu`).  An fp32 tensor
# may carry a plane TWIN (`t._hn_twin`: same values, both formats) -- layer outputs that are also returned to the caller.
PLANES = os.environ.get("HEATNET_PLANES", "1") != "0"  # A/B knob: 0 = fp32 tensors everywhere, operands split in the kernels


_last_fused = {}  # side channel from _ConvBnAct.forward to its wrapper (the output's scale slot and plane twin)


def is_planes(t):
    return getattr(t, "_hn_planes", None) is not None


class _PlanesToTensor(torch.autograd.Function):
    @staticmethod
    def forward(ctx, t):
        return planes_to_tensor(t)

    @staticmethod
    def backward(ctx, g):
        return g


def as_f32(t):
    """The activation as an ordinary fp32 tensor: a plane tensor is converted (one pass; gradients pass through unchanged)."""
    if t is not None and is_planes(t):
        slot = _plane_slot(t._hn_planes)
        out = _PlanesToTensor.apply(t)
        if slot is not None:
            out._hn_scale = (out._version, slot)
        return out
    return t


HN_Y_BF16 = 256  # include/heatnet_hip.h: OR-ed into `act`, the pre-normalisation tensor y holds bf16 elements
BF16_Y = os.environ.get("HEATNET_BF16_Y", "1") != "0"  # A/B knob of the bf16 arithmetic: 0 = y stays fp32 between conv and BatchNorm
BF16_PLANE = "bf16"  # `_hn_planes` marker of a ONE-plane bf16 tensor (the mixed-precision bf16 arithmetic): bf16(t), no scale


def _plane_slot(marker):
    """The scale slot a kernel takes for a plane tensor's marker (None: one bf16 plane)."""
    return None if isinstance(marker, str) else marker


def planes_of(t, bf16=False):
    """(plane tensor, marker) of an activation in the format of the current arithmetic (`bf16`: one bf16 plane; else fp16
    hi / lo planes with a scale slot = the marker): itself, its twin, or a twin made now (hn_absmax_scale unless the tensor
    carries a scale + hn_tensor_to_planes) and remembered on the tensor."""
    fits = lambda m: isinstance(m, str) == bf16
    marker = getattr(t, "_hn_planes", None)
    if marker is not None:
        if fits(marker):
            return t, marker
        t = as_f32(t)  # a plane tensor of the other arithmetic (the arithmetic was switched between two layers)
    twin = getattr(t, "_hn_twin", None)
    if twin is not None and twin[0] == t._version and fits(twin[1]._hn_planes):
        return twin[1], twin[1]._hn_planes
    x, ldx = as_nhwc(t)
    n, c, h, w = x.shape
    marker = BF16_PLANE if bf16 else tensor_scale(x, ldx, owner=t)
    pt = new_nhwc(n, c, h, w, x.device)
    check(_lib.load().hn_tensor_to_planes(_ptr(x), ldx, n * h * w, c, _ptr(_plane_slot(marker)), _ptr(pt), n * h * w * c, c,
                                          _stream()), "hn_tensor_to_planes")
    pt._hn_planes = marker
    t._hn_twin = (t._version, pt)
    return pt, marker


def planes_to_tensor(t):
    """fp32 values of a plane tensor (diagnostics / tests)."""
    slot = _plane_slot(t._hn_planes)
    n, c, h, w = t.shape
    out = new_nhwc(n, c, h, w, t.device)
    check(_lib.load().hn_planes_to_tensor(_ptr(t), n * h * w * c, c, _ptr(slot), _ptr(out), c, n * h * w, c, _stream()),
          "hn_planes_to_tensor")
    return out


def _plane_variants(cout, cred, taps):
    """Tiles of conv_igemm4 offered to the tuner for `cout` output channels, `cred` reduction channels and `taps` filter taps:
    0 = 128 x 128 (two workgroups per CU), 1 = 128 x 64, 2 = 128 x 256, 3 = 256 x 128 (eight waves, three stages), 7 = 224 x 256
    (eight waves, seven 16-row blocks per row wave: fewer operand bytes per multiply-add, and a tile HEIGHT that turns the 1.72 /
    3.44 rounds the 128-row tiles of the 40 x 88 and 80 x 176 layers make on 256 CUs into 0.98 / 1.97), 8 = 112 x 128 (four waves side by
    side, the same trick for the 512 slots of the two-workgroups-per-CU kernels); 4 = 128 x 64, 5 = 64 x 128, 6 = 64 x 64 with the DMA two / three K steps ahead -- for the layers with
    little matrix work per byte (1x1 convolutions, few channels), which otherwise wait out the memory latency every K step."""
    if cout <= 64:
        return [1, 4, 6]
    light = taps == 1 or cred <= 128
    return ([0, 1, 2, 3] + ([7] if (BIG_TILE and cout >= 256 and cred * taps >= 256) else []) +
            ([8] if (ROW7_TILE and cout >= 128) else []) + ([4, 5, 6] if light else []))


BIG_TILE = os.environ.get("HEATNET_BIG_TILE", "1") != "0"  # A/B knobs: offer the 224 x 256 tile (code 7) /
ROW7_TILE = os.environ.get("HEATNET_ROW7_TILE", "0") == "1"  # the 112 x 128 tile (code 8) to the tuner -- bitwise-tested, chosen by
# the isolated tuner for a dozen geometries, and the step 0.25 ms SLOWER with it (37.60 37.51 37.87 against 37.33 37.45 37.46,
# gpurun_out/r4/ab15): off


def _wgrad_plane_candidates(cout, cin, bf16=False):
    tms = (0, 1) if cout % 128 == 0 else (0,)
    tns = (0, 1) if cin % 128 == 0 else (0,)
    cands = []
    for tm in tms:
        for tn in tns:
            wide = cout % 128 == 0 and tm != 1 and cin % 128 == 0 and tn != 1
            # bk 2: f16x2 -- three LDS stages instead of two (not on the 128 x 128 tile); bf16 -- four instead of three
            for bk in ((0, 2) if (bf16 or not wide) else (0,)):
                cands += [(split, bk + 4 * tm + 16 * tn) for split in (0, -2, -3, -4)]
    if WGRAD_HUGE and not bf16 and cout % 256 == 0 and cin % 256 == 0:
        cands += [(split, 4 * 3 + 16 * 3) for split in (0, -2, -3, -4)]  # 256 x 256, eight waves (conv_wgrad4_kernel<4, 2, 2, 2, 2, 4>)
    return cands


# Offer the 256 x 256 weight-gradient tile to the tuner.  Built, bitwise-tested, taken by the tuner for the long 3x3 layers in
# isolation -- and the step measured 0.1 ms SLOWER with it (gpurun_out/r4/ab8: 38.12 38.01 38.20 against 37.94 37.94 38.15): off.
WGRAD_HUGE = os.environ.get("HEATNET_WGRAD_HUGE", "0") == "1"


def fused_conv_bn_ok(x, conv_w, stride, pad, dil, bn_training, shape=None):
    """Does the fused plane-tensor node cover this conv -> BatchNorm pair?  (f16x2 layer, batch statistics, whole 64-channel
    blocks both ways, enough pixels that neither the forward nor the data gradient would want split-K)  `shape`: the input's
    (N, C, H, W) when it differs from x's (a resampling in between)."""
    if not (PLANES and CONV_MATH in ("f16x2", "bf16") and not _MATH_F32_KINDS and bn_training and x.is_cuda and x.dim() == 4):
        return False
    cout, cin, kh, kw = conv_w.shape
    n, c_in, h, w = shape if shape is not None else x.shape
    if c_in != cin or cin % 64 != 0 or cout % 64 != 0 or _layer_math("fwd", cin, cout, kh, kw, conv_w) != CONV_MATH:
        return False
    ho, wo = conv_out_size(h, kh, stride, pad, dil), conv_out_size(w, kw, stride, pad, dil)
    if h * w < 128 or ho * wo < 128 or n * ho * wo <= 2048 or n * h * w <= 2048 * (4 if stride == 2 else 1):
        return False
    if stride not in (1, 2) or (stride == 2 and dil % 2 == 0):
        return False
    return n * h * w * cin * 2 < 0xFFFF0000 and n * ho * wo * cout * 2 < 0xFFFF0000 - 64 * cout * 2


BN_REDUCE_FUSION = os.environ.get("HEATNET_BN_REDUCE_FUSION", "1") != "0"  # A/B knob, see _BnLink
bn_reduce_counts = {"fused": 0, "discarded": 0, "standalone": 0}  # per process: how the BatchNorm backward reductions ran


class _BnLink:
    """Rides on the output of a fused conv -> BatchNorm (-> ReLU) node to the fused node that consumes it: the consumer's
    data-gradient launch produces exactly this layer's dz, so its epilogue takes the first pass of this layer's BatchNorm
    backward along (hn_conv2d_dgrad_planes_bnreduce) and the stand-alone reduction pass over dz and y is not run.  The
    producer's backward accepts the sums only if the gradient it receives IS the buffer that launch wrote (autograd hands a
    sole consumer's gradient through untouched; a sum of several is a new tensor) -- otherwise it clears them and reduces
    itself."""
    __slots__ = ("y", "ybf", "par", "rbits", "ws", "act", "cout", "npix", "fused_ptr")

    def __init__(self, y, ybf, par, rbits, ws, act, cout, npix):
        self.y, self.ybf, self.par, self.rbits, self.ws, self.act, self.cout, self.npix = y, ybf, par, rbits, ws, act, cout, npix
        self.fused_ptr = None


class _ConvBnAct(torch.autograd.Function):
    """z = act(bn(conv(x, w) + b) [+ residual]) [* drop_mask] as ONE autograd node; x (and the residual) may be plane tensors,
    z is written as fp32, as a plane tensor, or both; the gradient dy between the BatchNorm backward and the convolution's
    backward kernels never leaves the node and is a plane tensor.  cm/models/extractors.py:84-102, cm/models/pspnet.py:28-40."""

    @staticmethod
    def forward(ctx, x, w, b, gamma, beta, running_mean, running_var, residual, prelu_w, cfg):
        lib = _lib.load()
        (stride, pad, dil, act, leaky, momentum, eps, ws, drop_mask, passthrough, out_mode, mask_max) = cfg
        s = _stream()
        bf16 = CONV_MATH == "bf16"  # one bf16 plane per operand, no scales; else f16x2: fp16 hi / lo planes + scale slots
        xp, x_mark = planes_of(x, bf16)
        x_slot = _plane_slot(x_mark)
        n, cin, h, wd = xp.shape
        w_in, w = w, weight_ohwi(w)
        cout, _, kh, kw = w.shape
        ctx.in_dual_region = dual_region[0]
        in_link = getattr(x, "_hn_bn_link", None)
        ctx.in_link = in_link if (BN_REDUCE_FUSION and in_link is not None and stride == 1 and ctx.needs_input_grad[0]
                                  and in_link.cout == cin and in_link.npix == n * h * wd) else None
        ho, wo = conv_out_size(h, kh, stride, pad, dil), conv_out_size(wd, kw, stride, pad, dil)
        npix = n * ho * wo
        dev = xp.device
        wh, ctx.wt, ctx.wth = _weight_operands(w_in, w, stride, pad, dil, CONV_MATH, ctx.needs_input_grad[0], s)
        # one-plane bf16 arithmetic: the pre-normalisation tensor never leaves this node -- kept as bf16 (what the reference's
        # mixed-precision mode stores between its convolution and BatchNorm kernels, in fp16), statistics from the fp32 accumulators
        ybf = HN_Y_BF16 if (bf16 and BF16_Y and cout % 4 == 0 and out_mode != "f32") else 0  # (the plane-writing passes read it)
        if ybf:
            y = torch.empty((n, ho, wo, cout), device=dev, dtype=torch.bfloat16).permute(0, 3, 1, 2)
        else:
            y = new_nhwc(n, cout, ho, wo, dev)
        d = _desc(n, h, wd, cin, cin, ho, wo, cout, cout, kh, kw, stride, pad, dil)
        ws.open_stats()
        xplane = n * h * wd * cin
        if RANGE_CENSUS and not bf16:
            _census_operand(_leaf(w_in), 0, None, xp, cin, x_slot, n * h * wd, cin)
        tag = ".bf16p" if bf16 else ".planes"
        mcode = 3 if bf16 else 4

        def launch(v, st, keys):
            check(lib.hn_conv2d_fwd_planes(_ptr(xp), xplane, _ptr(x_slot), _ptr(wh), _ptr(b), _ptr(y), _ptr(st), _ptr(keys),
                                           ctypes.byref(d), ACT_NONE | ybf, 0.0, v, s), "hn_conv2d_fwd_planes")
        # (timed as it will run: with the statistics / range-key epilogue, into scratch copies of the accumulators)
        var = _pick("fwd" + tag + ".st", d, _plane_variants(cout, cin, kh * kw),
                    lambda v: launch(v, _tune_scratch(ws.stats), None if bf16 else _tune_scratch(ws.keys)))
        with _ConvTimer("fwd", d, lambda: _plan_name(0, d, 3, _variant(var & 3, var >> 2, 0, mcode))):
            launch(var, ws.stats, None if bf16 else ws.keys)
        # batch statistics -> normalisation constants, and the bound of |z| (the scale z is written with)
        par = torch.empty(5 * cout, device=dev, dtype=torch.float32)  # mean | invstd | scale | shift | max |xhat|
        mean, invstd, scale, shift, xhat = (par[i * cout:(i + 1) * cout] for i in range(5))
        res_p = res_f = None
        res_slot = None
        ldr = 0
        if residual is not None:
            if act in (ACT_PRELU, ACT_LEAKY):
                raise RuntimeError("conv_bn_act: a residual input is only supported with ReLU / no activation")
            if is_planes(residual) and isinstance(residual._hn_planes, str) == bf16:
                res_p, res_slot = residual, _plane_slot(residual._hn_planes)
            else:
                res_f, ldr = as_nhwc(as_f32(residual))
                res_slot = None if bf16 else tensor_scale(res_f, ldr, owner=residual)
        if bf16:
            z_slot = None
            check(lib.hn_bn_finalize(_ptr(ws.stats), npix, _ptr(gamma), _ptr(beta), _ptr(running_mean), _ptr(running_var),
                                     momentum, eps, _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift), cout, s), "hn_bn_finalize")
        else:
            z_slot = _new_scale_slot(dev)
            check(lib.hn_bn_finalize_range(_ptr(ws.stats), _ptr(ws.keys), npix, _ptr(gamma), _ptr(beta), _ptr(running_mean),
                                           _ptr(running_var), momentum, eps, _ptr(mean), _ptr(invstd), _ptr(scale), _ptr(shift),
                                           _ptr(xhat), cout, act, _ptr(prelu_w), leaky, _ptr(res_slot), mask_max, _ptr(z_slot),
                                           s), "hn_bn_finalize_range")
        ws.stats_open = False
        zf = new_nhwc(n, cout, ho, wo, dev) if out_mode != "planes" else None
        zp = new_nhwc(n, cout, ho, wo, dev) if out_mode != "f32" else None
        if drop_mask is not None:
            drop_mask = drop_mask.contiguous()
        rbits = None
        if act == ACT_RELU and residual is not None and RELU_BITS and any(ctx.needs_input_grad):
            rbits = torch.empty(npix * (cout // 4), device=dev, dtype=torch.uint8)
        if zp is not None:
            check(lib.hn_bn_act_fwd_planes(_ptr(y), cout, _ptr(scale), _ptr(shift), _ptr(res_f), ldr, _ptr(res_p), npix * cout, cout,
                                           _ptr(res_slot) if res_p is not None else None, _ptr(zf), cout, _ptr(zp), npix * cout,
                                           cout, _ptr(z_slot), npix, cout, act | ybf, _ptr(prelu_w), leaky, _ptr(drop_mask), ho * wo,
                                           _ptr(rbits), s), "hn_bn_act_fwd_planes")
            zp._hn_planes = BF16_PLANE if bf16 else z_slot
        else:
            if res_p is not None:
                raise RuntimeError("conv_bn_act: a plane-tensor residual needs a plane (or dual) output")
            check(lib.hn_bn_act_fwd(_ptr(y), cout, _ptr(scale), _ptr(shift), _ptr(res_f), ldr, _ptr(zf), cout, npix, cout, act,
                                    _ptr(prelu_w), leaky, _ptr(drop_mask), ho * wo, None, _ptr(rbits), s), "hn_bn_act_fwd")
        ctx.cfg = (stride, pad, dil, act, leaky, passthrough, residual is not None, out_mode)
        ctx.ws, ctx.drop_mask, ctx.x_slot, ctx.bf16, ctx.ybf = ws, drop_mask, x_slot, bf16, ybf
        ctx.leaves = (_leaf(w_in), _leaf(b), _leaf(gamma), _leaf(beta), _leaf(prelu_w))
        # z itself is needed by the backward only for a ReLU behind a residual add without the byte mask
        keep_z = zf if (act == ACT_RELU and residual is not None and rbits is None and zf is not None) else None
        if act == ACT_RELU and residual is not None and rbits is None and zf is None and any(ctx.needs_input_grad):
            raise RuntimeError("conv_bn_act: plane-only output behind a residual ReLU needs the byte mask (C % 4, HEATNET_RELU_BITS)")
        ctx.save_for_backward(xp, w, y, par, gamma, prelu_w, rbits, keep_z)
        out = zf if zf is not None else zp
        _last_fused["slot"], _last_fused["twin"] = (BF16_PLANE if bf16 else z_slot), (zp if zf is not None else None)
        # what a consuming fused node needs to take this layer's backward reduction into its data-gradient epilogue
        ctx.link = None
        if (BN_REDUCE_FUSION and act in (ACT_RELU, ACT_NONE) and drop_mask is None and cout % 4 == 0 and any(ctx.needs_input_grad)
                and (residual is None or rbits is not None) and keep_z is None):
            ctx.link = _BnLink(y, bool(ybf), par, rbits, ws, act, cout, npix)
        _last_fused["link"] = ctx.link
        if passthrough:
            return out, x.view_as(x)
        return out

    @staticmethod
    def backward(ctx, dz, *extra):
        lib = _lib.load()
        xp, w, y, par, gamma, prelu_w, rbits, z = ctx.saved_tensors
        note_backward_stream(xp.device)
        stride, pad, dil, act, leaky, passthrough, has_res, out_mode = ctx.cfg
        dskip = extra[0] if (passthrough and extra) else None
        s = _stream()
        n, cin, h, wd = xp.shape
        cout, _, kh, kw = w.shape
        _, _, ho, wo = y.shape
        npix = n * ho * wo
        dev = y.device
        dz, lddz = as_nhwc(dz)
        if rbits is not None or ctx.ybf:
            dz, lddz = _vector_grad(dz, lddz)
        mean, invstd, scale, shift, xhat = (par[i * cout:(i + 1) * cout] for i in range(5))
        ws = ctx.ws
        link = ctx.link
        fused_reduce = False
        if link is not None and link.fused_ptr is not None:
            # the consumer's data-gradient epilogue already summed du / du * xhat of THIS dz (see _BnLink)?
            fused_reduce = ws.fused_owner is link and link.fused_ptr == dz.data_ptr() and lddz == cout
            if ws.fused_owner is link:
                ws.fused_owner = None
                if not fused_reduce:  # they belong to a partial gradient: start over
                    ws.bwd.zero_()
                    ws.gmax.zero_()
                    ws.bwd_open = False
                    ws.fuse_refused = True  # this layer's output has several readers: do not try again
                    bn_reduce_counts["discarded"] += 1
            link.fused_ptr = None
        bn_reduce_counts["fused" if fused_reduce else "standalone"] += 1
        if not fused_reduce:
            ws.open_bwd()
        lw, lbias, lg, lb, lp = ctx.leaves
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        # ---- BatchNorm backward: reductions (+ the bound of |dy|), then dy as a plane tensor ----
        dpar = torch.empty(2 * cout + 1, device=dev, dtype=torch.float32)
        sg = grad_slot(lg) if (lg is not None and ctx.needs_input_grad[3]) else None
        sb = grad_slot(lb) if (lb is not None and ctx.needs_input_grad[4]) else None
        sp = grad_slot(lp) if (lp is not None and prelu_w is not None and ctx.needs_input_grad[8]) else None
        used = [t for t in (sg, sb, sp) if t is not None]
        if len({first for _, first in used}) > 1:
            sg = sb = sp = None
            used = []
        accumulate = bool(used) and not used[0][1]
        bf16 = ctx.bf16
        if fused_reduce:
            dy_slot = None if bf16 else _new_scale_slot(dev)
            check(lib.hn_bn_act_bwd_finish(_ptr(ws.bwd), _ptr(dpar), cout, _ptr(sg[0]) if sg else None, _ptr(sb[0]) if sb else None,
                                           None, int(accumulate), None if bf16 else _ptr(ws.gmax), _ptr(gamma), _ptr(invstd),
                                           _ptr(xhat), npix, 1, _ptr(dy_slot), s), "hn_bn_act_bwd_finish")
        elif bf16:
            dy_slot = None
            check(lib.hn_bn_act_bwd_reduce(_ptr(dz), lddz, _ptr(z), cout, _ptr(y), cout, _ptr(mean), _ptr(invstd), _ptr(scale),
                                           _ptr(shift), npix, cout, act | ctx.ybf, _ptr(prelu_w), leaky, _ptr(ws.bwd), _ptr(dpar),
                                           _ptr(sg[0]) if sg else None, _ptr(sb[0]) if sb else None,
                                           _ptr(sp[0]) if sp else None, int(accumulate), _ptr(ctx.drop_mask), ho * wo,
                                           _ptr(rbits), s), "hn_bn_act_bwd_reduce")
        else:
            dy_slot = _new_scale_slot(dev)
            check(lib.hn_bn_act_bwd_reduce_range(_ptr(dz), lddz, _ptr(z), cout, _ptr(y), cout, _ptr(mean), _ptr(invstd),
                                                 _ptr(scale), _ptr(shift), npix, cout, act, _ptr(prelu_w), leaky, _ptr(ws.bwd),
                                                 _ptr(dpar), _ptr(sg[0]) if sg else None, _ptr(sb[0]) if sb else None,
                                                 _ptr(sp[0]) if sp else None, int(accumulate), _ptr(ctx.drop_mask), ho * wo,
                                                 _ptr(rbits), _ptr(ws.gmax), _ptr(gamma), _ptr(xhat), npix, 1, _ptr(dy_slot), s),
                  "hn_bn_act_bwd_reduce_range")
        ws.bwd_open = False
        dres = new_nhwc(n, cout, ho, wo, dev) if (has_res and ctx.needs_input_grad[7]) else None
        dx = dw_out = db = None
        if need_x or need_w or dres is not None:
            dyp = new_nhwc(n, cout, ho, wo, dev)
            check(lib.hn_bn_act_bwd_apply_planes(_ptr(dz), lddz, _ptr(z), cout, _ptr(y), cout, _ptr(mean), _ptr(invstd), _ptr(gamma),
                                                 _ptr(scale), _ptr(shift), _ptr(dpar), npix, 1, _ptr(dyp), npix * cout, cout,
                                                 _ptr(dy_slot), _ptr(dres), cout, npix, cout, act | ctx.ybf, _ptr(prelu_w), leaky,
                                                 _ptr(ctx.drop_mask), ho * wo, _ptr(rbits), s), "hn_bn_act_bwd_apply_planes")
        d = _desc(n, h, wd, cin, cin, ho, wo, cout, cout, kh, kw, stride, pad, dil)
        side = None
        # Only when the gradient goes into the parameter's arena slot: then nothing on the main stream touches it before the
        # end-of-backward join (a second use of the layer adds in place on the same side stream).  A free-standing dW is
        # handed to autograd, which may add to it on the main stream at once (a layer applied to the day and the night batch).
        # (not next to the two encoder streams of a late-fusion network, ops.DUAL_STREAMS: three compute streams beside the
        # communication stream and RCCL's own cost 10 ms per step in the N > 1 path, and two already fill the bubbles)
        if (WGRAD_STREAM and need_w and _conv_prof is None and not DETERMINISTIC and getattr(lw, "_hn_grad_slot", None) is not None
                and dev not in dual_stream_used):
            main = torch.cuda.current_stream(dev)
            ready = torch.cuda.Event()
            ready.record(main)  # dy (and, far earlier, the cleared gradient arena) is complete up to here; dgrad comes after
            side = _wgrad_side_stream(dev, main)
        elif (SECOND_STREAM_WGRAD and WGRAD_STREAM and need_w and _conv_prof is None and not DETERMINISTIC
              and getattr(lw, "_hn_grad_slot", None) is not None and dev in dual_stream_used and not ctx.in_dual_region
              and torch.cuda.current_stream(dev) != _dual_streams.get(dev)):
            # EXPERIMENT: behind the fusion point the second encoder stream is idle -- the weight gradients of the single-chain part
            # (layer3, layer4, decoder) run there, still two compute streams in all
            main = torch.cuda.current_stream(dev)
            ready = torch.cuda.Event()
            ready.record(main)
            side = _dual_streams[dev]
            _queue_second_stream_join(dev, main, side)
        if RANGE_CENSUS and (need_x or need_w) and not bf16:
            _census_operand(lw, 1, None, dyp, cout, dy_slot, npix, cout)
        tag = ".bf16p" if bf16 else ".planes"
        mcode = 3 if bf16 else 4
        # ---- data gradient ----
        if need_x:
            dx = new_nhwc(n, cin, h, wd, dev)
            add, ldadd = (None, 0) if dskip is None else as_nhwc(dskip)

            def launch(v, ad, ld, bn=None):
                if bn is not None:
                    rc = lib.hn_conv2d_dgrad_planes_bnreduce(_ptr(dyp), npix * cout, _ptr(dy_slot), _ptr(ctx.wth), _ptr(dx), _ptr(ad),
                                                             ld, ctypes.byref(d), v, ctypes.byref(bn), s)
                    if rc != _lib.HN_ESHAPE:
                        check(rc, "hn_conv2d_dgrad_planes_bnreduce")
                        return True
                check(lib.hn_conv2d_dgrad_planes(_ptr(dyp), npix * cout, _ptr(dy_slot), _ptr(ctx.wth), _ptr(dx), _ptr(ad), ld,
                                                 ctypes.byref(d), v, s), "hn_conv2d_dgrad_planes")
                return False
            # dx is the dz of the BatchNorm layer that produced this node's input: its backward reduction rides along
            il, bn_args, tune_bn = ctx.in_link, None, None
            if il is not None and il.ws.fused_owner is None and not il.ws.bwd_open and not il.ws.fuse_refused:
                lmean, linvstd, lscale, lshift = (il.par[i * cin:(i + 1) * cin] for i in range(4))
                bn_args = _lib.HnBnReduceArgs(_ptr(il.y), cin, int(il.ybf), il.act, 0, _ptr(lscale), _ptr(lshift), _ptr(lmean),
                                              _ptr(linvstd), _ptr(il.rbits), _ptr(il.ws.bwd), None if bf16 else _ptr(il.ws.gmax))
                # the tuner times the launch AS IT WILL RUN -- the instantiation with the reduction (its own registers and loads)
                # and the addend -- with the sums going to a scratch copy
                def tune_bn():
                    return _lib.HnBnReduceArgs(_ptr(il.y), cin, int(il.ybf), il.act, 0, _ptr(lscale), _ptr(lshift), _ptr(lmean),
                                               _ptr(linvstd), _ptr(il.rbits), _ptr(_tune_scratch(il.ws.bwd)),
                                               None if bf16 else _ptr(_tune_scratch(il.ws.gmax)))
            var = _pick("dgrad" + tag + (".br" if bn_args is not None else "") + (".add" if add is not None else ""), d,
                        _plane_variants(cin, cout, kh * kw), lambda v: launch(v, add, ldadd, tune_bn() if tune_bn else None))
            extra = bn_extra = 4.0 * n * h * wd * cin if add is not None else 0.0
            if bn_args is not None:  # y of the producing layer (fp32 or bf16) and its byte masks, read by the epilogue
                bn_extra = extra + ((2.0 if il.ybf else 4.0) + (1.0 if il.rbits is not None else 0.0)) * n * h * wd * cin
            with _ConvTimer("dgrad", d, lambda: _plan_name(1, d, 3, _variant(var & 3, var >> 2, 0, mcode)), extra) as tm:
                if launch(var, add, ldadd, bn_args):
                    if tm.on:
                        tm.bytes += bn_extra - extra
                        tm.kname = _plan_name(1, d, 7, _variant(var & 3, var >> 2, 0, mcode))  # (the instantiation with the reduction)
                    il.ws.fused_owner, il.ws.bwd_open, il.fused_ptr = il, True, dx.data_ptr()
        elif dskip is not None:
            dx = dskip
        # ---- weight gradient ----
        if need_w:
            slot = grad_slot(lw) if lw is not None else None
            flags = 0
            if slot is not None and weight_ohwi(slot[0]) is slot[0]:
                dw = slot[0]
                flags = 1 if slot[1] else 2
                dw_out = dw if slot[1] else None
            else:
                dw_out = dw = torch.empty((cout, kh, kw, cin), device=dev, dtype=torch.float32).permute(0, 3, 1, 2)
            xplane = n * h * wd * cin

            def wgrad(dst, sp_, v, fl, st=s):
                check(lib.hn_conv2d_wgrad_planes(_ptr(xp), xplane, _ptr(ctx.x_slot), _ptr(dyp), npix * cout, _ptr(dy_slot), _ptr(dst),
                                                 ctypes.byref(d), sp_, v, fl, st), "hn_conv2d_wgrad_planes")
            if DETERMINISTIC:
                splits = max(1, min(32, npix // 4096))
                wsp = _det_workspace(dev, splits * dw.numel())
                with _ConvTimer("wgrad", d, lambda: _wgrad_plan_name(d, 2, _variant(0, 0, 0, mcode))):
                    check(lib.hn_conv2d_wgrad_det_planes(_ptr(xp), xplane, _ptr(ctx.x_slot), _ptr(dyp), npix * cout, _ptr(dy_slot),
                                                         _ptr(dw), ctypes.byref(d), splits, 0, flags & 2, _ptr(wsp), s),
                          "hn_conv2d_wgrad_det_planes")
            else:
                split, var = _pick("wgrad" + tag, d, _wgrad_plane_candidates(cout, cin, bf16),
                                   lambda sv: wgrad(_tune_scratch(dw) if flags else dw, sv[0], sv[1], 0))
                if side is not None and slot is not None and dw is slot[0]:
                    # beside the main stream's next BatchNorm backward passes (see WGRAD_STREAM); the operands stay reserved
                    # for the side stream until it is done with them
                    side.wait_event(ready)
                    wgrad(dw, split, var, flags, ctypes.c_void_p(side.cuda_stream))
                    for t in (xp, dyp, dw, ctx.x_slot, dy_slot):
                        if t is not None:
                            t.record_stream(side)
                else:
                    with _ConvTimer("wgrad", d, lambda: _wgrad_plan_name(d, 2, var + 64 * mcode)):
                        wgrad(dw, split, var, flags)
        if lbias is not None and ctx.needs_input_grad[2]:
            # the output is normalised with batch statistics: the bias cancels, its gradient is exactly zero
            bslot = grad_slot(lbias, accumulate_ok=False)
            db = bslot[0] if bslot is not None else torch.zeros(cout, device=dev, dtype=torch.float32)
        elif ctx.needs_input_grad[2]:
            db = torch.zeros(cout, device=dev, dtype=torch.float32)

        def out(slot_, fallback, needed):
            if not needed:
                return None
            if slot_ is None:
                return fallback
            return slot_[0] if slot_[1] else None
        dgamma = out(sg, dpar[cout:2 * cout], ctx.needs_input_grad[3])
        dbeta = out(sb, dpar[0:cout], ctx.needs_input_grad[4])
        dprelu = out(sp, dpar[2 * cout:2 * cout + 1], prelu_w is not None and ctx.needs_input_grad[8])
        return dx, dw_out, db, dgamma, dbeta, None, None, dres, dprelu, None


def conv_bn_act_fused(x, w, b, gamma, beta, running_mean, running_var, residual, prelu_w, stride, pad, dil, act, leaky, momentum,
                      eps, ws, drop_mask, passthrough, out_mode, mask_max):
    """See _ConvBnAct.  out_mode: "f32" | "planes" | "dual" (fp32 tensor carrying its plane twin)."""
    cfg = (int(stride), int(pad), int(dil), int(act), float(leaky), float(momentum), float(eps), ws, drop_mask, bool(passthrough),
           out_mode, float(mask_max))
    out = _ConvBnAct.apply(x, w, b, gamma, beta, running_mean, running_var, residual, prelu_w, cfg)
    z, skip = out if passthrough else (out, None)
    slot, twin = _last_fused.pop("slot"), _last_fused.pop("twin")
    link = _last_fused.pop("link", None)
    if link is not None:
        z._hn_bn_link = link  # (NOT handed to the pass-through alias: one consumer takes the reduction)
    if out_mode == "planes":
        z._hn_planes = slot
    else:
        if not isinstance(slot, str):
            z._hn_scale = (z._version, slot)  # (a bound of max |z|: valid wherever a scale is taken)
        if twin is not None:
            z._hn_twin = (z._version, twin)
    if passthrough:
        if is_planes(x):
            skip._hn_planes = x._hn_planes
        else:
            tw = getattr(x, "_hn_twin", None)
            if tw is not None:
                skip._hn_twin = (skip._version, tw[1])
            _inherit_scale(skip, x)
        return z, skip
    return z

_in_consts = {}


def instance_norm(x, eps=1e-5):
    """nn.InstanceNorm2d (no affine, no running statistics): every image normalised with its own per-channel statistics
    -- the BatchNorm kernels run once per image with unit scale / zero shift (an optional branch: the few small launches
    per image do not matter, cm/models/downscale_network.py:18)."""
    n, c = x.shape[0], x.shape[1]
    key = (x.device, c)
    if key not in _in_consts:
        _in_consts[key] = (torch.ones(c, device=x.device), torch.zeros(c, device=x.device))
    one, zero = _in_consts[key]
    outs = [bn_act(x[i:i + 1], None, one, zero, None, None, training=True, eps=eps) for i in range(n)]
    return outs[0] if n == 1 else concat_batch(outs)


def concat_batch(xs):
    """torch.cat(xs, 0) for NHWC tensors (a plain copy; autograd hands the slices back)."""
    return torch.cat([as_nhwc(t)[0] for t in xs], 0)


def drop_mask(n, c, p, device):
    """nn.Dropout2d's per-(image, channel) factor: 0 with probability p, else 1/(1-p).  Drawn by hn_dropout_mask from a
    64-bit seed taken from torch's CPU generator (so torch.manual_seed reproduces a run)."""
    mask = torch.empty((n, c), device=device, dtype=torch.float32)
    seed = int(torch.empty((), dtype=torch.int64).random_().item())
    check(_lib.load().hn_dropout_mask(_ptr(mask), n * c, float(p), ctypes.c_uint64(seed & 0xFFFFFFFFFFFFFFFF), _stream()),
          "hn_dropout_mask")
    return mask


# ------------------------------------------------------------------------------------------------------
# spatial operators
# ------------------------------------------------------------------------------------------------------

class _MaxPool3x3s2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        n, c, h, w = x.shape
        ho, wo = (h - 1) // 2 + 1, (w - 1) // 2 + 1
        y = new_nhwc(n, c, ho, wo, x.device)
        idx = torch.empty(n * ho * wo * c, device=x.device, dtype=torch.uint8)
        check(lib.hn_maxpool3x3s2_fwd(_ptr(x), ldx, _ptr(y), c, _ptr(idx), n, h, w, c, ho, wo, _stream()),
              "hn_maxpool3x3s2_fwd")
        ctx.shape = (n, c, h, w, ho, wo)
        ctx.save_for_backward(idx)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (idx,) = ctx.saved_tensors
        n, c, h, w, ho, wo = ctx.shape
        dy, lddy = as_nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy.device)
        check(lib.hn_maxpool3x3s2_bwd(_ptr(dy), lddy, _ptr(idx), _ptr(dx), c, n, h, w, c, ho, wo, _stream()),
              "hn_maxpool3x3s2_bwd")
        return dx


def maxpool3x3s2(x):
    return _inherit_scale(_MaxPool3x3s2.apply(x), x)


class _AdaptiveAvgPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, ho, wo, x.device)
        check(lib.hn_adaptive_avgpool_fwd(_ptr(x), ldx, _ptr(y), c, n, h, w, c, ho, wo, _stream()),
              "hn_adaptive_avgpool_fwd")
        ctx.shape = (n, c, h, w, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        n, c, h, w, ho, wo = ctx.shape
        dy, lddy = as_nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy.device)
        check(lib.hn_adaptive_avgpool_bwd(_ptr(dy), lddy, _ptr(dx), c, n, h, w, c, ho, wo, _stream()),
              "hn_adaptive_avgpool_bwd")
        return dx, None, None


def adaptive_avg_pool(x, size):
    ho, wo = (size, size) if isinstance(size, int) else size
    return _inherit_scale(_AdaptiveAvgPool.apply(x, int(ho), int(wo)), x)


class _PyramidPool(torch.autograd.Function):
    """The adaptive average poolings of all pyramid levels of one map; their gradients return to the map in one
    kernel (hn_pyramid_pool_bwd) instead of one full-size tensor per level plus autograd's adds."""

    @staticmethod
    def forward(ctx, x, *sizes):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        n, c, h, w = x.shape
        outs = []
        k = len(sizes)
        if PYRAMID_FWD_FUSED and 0 < k <= 8 and all(ho <= h and wo <= w for ho, wo in sizes):
            # one read of the map for every level (hn_pyramid_pool_fwd) instead of one per level
            outs = [new_nhwc(n, c, ho, wo, x.device) for ho, wo in sizes]
            ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in outs])
            hos = (ctypes.c_int * k)(*[ho for ho, _ in sizes])
            wos = (ctypes.c_int * k)(*[wo for _, wo in sizes])
            lds = (ctypes.c_int * k)(*[c] * k)
            nseg = lib.hn_pyramid_merge_bwd_segments(n, h, w)
            wsp = torch.empty(n * h * nseg * sum(wo for _, wo in sizes) * c, device=x.device, dtype=torch.float32)
            check(lib.hn_pyramid_pool_fwd(_ptr(x), ldx, k, ptrs, hos, wos, lds, _ptr(wsp), n, h, w, c, _stream()), "hn_pyramid_pool_fwd")
        else:
            for ho, wo in sizes:
                y = new_nhwc(n, c, ho, wo, x.device)
                check(lib.hn_adaptive_avgpool_fwd(_ptr(x), ldx, _ptr(y), c, n, h, w, c, ho, wo, _stream()),
                      "hn_adaptive_avgpool_fwd")
                outs.append(y)
        ctx.shape = (n, c, h, w)
        ctx.sizes = sizes
        ctx.set_materialize_grads(False)  # a level nobody differentiates through is skipped, not zero-filled
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        lib = _lib.load()
        n, c, h, w = ctx.shape
        live = [(as_nhwc(dy), sz) for dy, sz in zip(dys, ctx.sizes) if dy is not None]
        dx = new_nhwc(n, c, h, w, live[0][0][0].device)
        k = len(live)
        ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for (t, _), _ in live])
        hos = (ctypes.c_int * k)(*[sz[0] for _, sz in live])
        wos = (ctypes.c_int * k)(*[sz[1] for _, sz in live])
        lds = (ctypes.c_int * k)(*[ld for (_, ld), _ in live])
        check(lib.hn_pyramid_pool_bwd(k, ptrs, hos, wos, lds, _ptr(dx), c, n, h, w, c, _stream()),
              "hn_pyramid_pool_bwd")
        return (dx,) + (None,) * len(ctx.sizes)


def pyramid_pool(x, sizes):
    """[adaptive_avg_pool(x, s) for s in sizes] with a single fused backward."""
    sizes = tuple((int(s), int(s)) if isinstance(s, int) else (int(s[0]), int(s[1])) for s in sizes)
    outs = _PyramidPool.apply(x, *sizes)
    for o in outs:
        _inherit_scale(o, x)  # averages of x's elements
    return outs


class _Bilinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ho, wo):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, ho, wo, x.device)
        check(lib.hn_bilinear_fwd(_ptr(x), ldx, _ptr(y), c, n, h, w, c, ho, wo, _stream()), "hn_bilinear_fwd")
        ctx.shape = (n, c, h, w, ho, wo)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        n, c, h, w, ho, wo = ctx.shape
        dy, lddy = as_nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy.device)
        check(lib.hn_bilinear_bwd(_ptr(dy), lddy, _ptr(dx), c, n, h, w, c, ho, wo, _stream()), "hn_bilinear_bwd")
        return dx, None, None


class _BilinearUp2Planes(torch.autograd.Function):
    """F.upsample(x2, bilinear) written as a plane tensor with the input's scale slot (cm/models/pspnet.py:36-37: the
    upsampled map is read by the block's convolution only)."""

    @staticmethod
    def forward(ctx, x, slot):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        n, c, h, w = x.shape
        y = new_nhwc(n, c, 2 * h, 2 * w, x.device)
        check(lib.hn_bilinear_up2_planes(_ptr(x), ldx, _ptr(slot), _ptr(y), n * 4 * h * w * c, c, n, h, w, c, _stream()),
              "hn_bilinear_up2_planes")
        ctx.shape = (n, c, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        n, c, h, w = ctx.shape
        dy, lddy = as_nhwc(dy)
        dx = new_nhwc(n, c, h, w, dy.device)
        check(lib.hn_bilinear_bwd(_ptr(dy), lddy, _ptr(dx), c, n, h, w, c, 2 * h, 2 * w, _stream()), "hn_bilinear_bwd")
        return dx, None


def bilinear_up2_planes(x):
    """Exact x2 bilinear upsampling as a plane tensor (for conv_bn_act only)."""
    x = as_f32(x)
    if CONV_MATH == "bf16":
        y = _BilinearUp2Planes.apply(x, None)
        y._hn_planes = BF16_PLANE
        return y
    xn, ldx = as_nhwc(x)
    slot = tensor_scale(xn, ldx, owner=x)
    y = _BilinearUp2Planes.apply(x, slot)
    y._hn_planes = slot
    return y


def bilinear(x, size):
    if (CONV_MATH == "f16x2" and x.is_cuda and x.dim() == 4 and getattr(x, "_hn_scale", None) is None and
            int(size[0]) * int(size[1]) > x.shape[2] * x.shape[3] and x.shape[1] % 32 == 0):
        tensor_scale(*as_nhwc(x), owner=x)  # an upsampling: find the scale on the small tensor, the large one inherits it
    return _inherit_scale(_Bilinear.apply(x, int(size[0]), int(size[1])), x)


PYRAMID_BWD_FUSED = os.environ.get("HEATNET_PYRAMID_BWD_FUSED", "1") != "0"  # A/B knobs: the pyramid merge's backward /
# ... the pyramid poolings from one read of the map (hn_pyramid_pool_fwd: built, tested, the step measured the same with it --
# 37.67 37.66 37.80 against 37.63 37.78 37.66 ms, gpurun_out/r4/ab12 -- the block-per-bin kernels are not the problem: off)
PYRAMID_FWD_FUSED = os.environ.get("HEATNET_PYRAMID_FWD_FUSED", "0") == "1"


class _PyramidMerge(torch.autograd.Function):
    """relu(main + sum_i bilinear_up(prior_i)): see hn_pyramid_merge_fwd."""

    @staticmethod
    def forward(ctx, main, *priors):
        lib = _lib.load()
        main, ldm = as_nhwc(main)
        n, c, h, w = main.shape
        ps = [as_nhwc(p) for p in priors]
        k = len(ps)
        ptrs = (ctypes.c_void_p * k)(*[p.data_ptr() for p, _ in ps])
        hs = (ctypes.c_int * k)(*[p.shape[2] for p, _ in ps])
        ws = (ctypes.c_int * k)(*[p.shape[3] for p, _ in ps])
        lds = (ctypes.c_int * k)(*[ld for _, ld in ps])
        out = new_nhwc(n, c, h, w, main.device)
        check(lib.hn_pyramid_merge_fwd(_ptr(main), ldm, k, ptrs, hs, ws, lds, _ptr(out), c, n, h, w, c, 1, _stream()),
              "hn_pyramid_merge_fwd")
        ctx.sizes = [(p.shape[2], p.shape[3]) for p, _ in ps]
        ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        (out,) = ctx.saved_tensors
        dout, ldd = as_nhwc(dout)
        n, c, h, w = out.shape
        s = _stream()
        du = new_nhwc(n, c, h, w, out.device)
        k = len(ctx.sizes)
        if (PYRAMID_BWD_FUSED and k > 0 and all(ctx.needs_input_grad[1:1 + k]) and all(ph <= h and pw <= w for ph, pw in ctx.sizes)):
            # one read of dout for du AND every prior's gradient (hn_pyramid_merge_bwd): level by level, hn_bilinear_bwd read the
            # 115 MB of du four more times at a quarter of the HBM rate
            dps = [new_nhwc(n, c, ph, pw, out.device) for ph, pw in ctx.sizes]
            ptrs = (ctypes.c_void_p * k)(*[t.data_ptr() for t in dps])
            hs = (ctypes.c_int * k)(*[ph for ph, _ in ctx.sizes])
            ws = (ctypes.c_int * k)(*[pw for _, pw in ctx.sizes])
            lds = (ctypes.c_int * k)(*[c] * k)
            nseg = lib.hn_pyramid_merge_bwd_segments(n, h, w)
            wsp = torch.empty(n * h * nseg * sum(pw for _, pw in ctx.sizes) * c, device=out.device, dtype=torch.float32)
            check(lib.hn_pyramid_merge_bwd(_ptr(dout), ldd, _ptr(out), c, _ptr(du), c, k, ptrs, hs, ws, lds, _ptr(wsp), n, h, w, c, s),
                  "hn_pyramid_merge_bwd")
            return (du,) + tuple(dps)
        check(lib.hn_act_bwd(_ptr(dout), ldd, _ptr(out), c, _ptr(du), c, n * h * w, c, ACT_RELU, 0.0, s), "hn_act_bwd")
        grads = [du]
        for i, (ph, pw) in enumerate(ctx.sizes):
            if not ctx.needs_input_grad[1 + i]:
                grads.append(None)
                continue
            dp = new_nhwc(n, c, ph, pw, out.device)
            check(lib.hn_bilinear_bwd(_ptr(du), c, _ptr(dp), c, n, ph, pw, c, h, w, s), "hn_bilinear_bwd")
            grads.append(dp)
        return tuple(grads)


def pyramid_merge(main, priors):
    return _PyramidMerge.apply(main, *priors)


class _ScaleNC(torch.autograd.Function):
    """x * mask[n, c] -- Dropout2d with a pre-drawn keep/(1-p) mask."""

    @staticmethod
    def forward(ctx, x, mask):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        n, c, h, w = x.shape
        mask = mask.contiguous()
        y = new_nhwc(n, c, h, w, x.device)
        check(lib.hn_scale_nc(_ptr(x), ldx, _ptr(mask), _ptr(y), c, n, h * w, c, _stream()), "hn_scale_nc")
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (mask,) = ctx.saved_tensors
        dy, lddy = as_nhwc(dy)
        n, c, h, w = dy.shape
        dx = new_nhwc(n, c, h, w, dy.device)
        check(lib.hn_scale_nc(_ptr(dy), lddy, _ptr(mask), _ptr(dx), c, n, h * w, c, _stream()), "hn_scale_nc")
        return dx, None


def dropout2d(x, p, training, mask=None):
    """nn.Dropout2d: whole (n, c) maps dropped with probability p and survivors scaled by 1/(1-p)."""
    if mask is None:
        if not training or p == 0.0:
            return x
        mask = drop_mask(x.shape[0], x.shape[1], p, x.device)
    return _ScaleNC.apply(x, mask)


class _Concat(torch.autograd.Function):
    """torch.cat(dim=1) as strided copies into one NHWC buffer; backward hands out channel-slice views."""

    @staticmethod
    def forward(ctx, *xs):
        lib = _lib.load()
        xs = [as_nhwc(x) for x in xs]
        n, _, h, w = xs[0][0].shape
        ctot = sum(x.shape[1] for x, _ in xs)
        out = new_nhwc(n, ctot, h, w, xs[0][0].device)
        c0, s = 0, _stream()
        ctx.splits = []
        for x, ld in xs:
            c = x.shape[1]
            check(lib.hn_copy(_ptr(x), ld, ctypes.c_void_p(out.data_ptr() + 4 * c0), ctot, n * h * w, c, s), "hn_copy")
            ctx.splits.append((c0, c))
            c0 += c
        return out

    @staticmethod
    def backward(ctx, dy):
        return tuple(dy[:, c0:c0 + c] for c0, c in ctx.splits)


class _SplitColumns(torch.autograd.Function):
    """w[:, a_i:b_i] for consecutive input-channel ranges of a 1x1 weight (the pyramid bottleneck applied per pyramid level,
    pspnet.PSPModule) as ONE autograd node: the backward places every slice's gradient into its columns of one full-size
    gradient -- the parameter's arena slot when it has one -- with strided copies.  Plain slicing costs, per slice, a
    full-size zero fill, a copy and a full-size accumulation add in ATen."""

    @staticmethod
    def forward(ctx, w, *sizes):
        ctx.sizes = sizes
        ctx.leaf = _leaf(w)
        ctx.shape, ctx.strides = tuple(w.shape), tuple(w.stride())
        outs, a = [], 0
        for f in sizes:
            outs.append(w[:, a:a + f])
            a += f
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        lib = _lib.load()
        cout, ctot = ctx.shape[0], ctx.shape[1]
        slot = grad_slot(ctx.leaf, accumulate_ok=False) if ctx.leaf is not None else None
        dev = next(g.device for g in grads if g is not None)
        if slot is not None and slot[1] and slot[0].stride() == ctx.strides:
            full = slot[0]
        else:
            full = torch.empty_strided(ctx.shape, ctx.strides, device=dev, dtype=torch.float32)
        dense = full.stride(1) == 1 and full.stride(0) == ctot  # [Cout][Ctot] rows (OIHW 1x1 in channels-last memory)
        a = 0
        for f, g in zip(ctx.sizes, grads):
            dst = full[:, a:a + f]
            if g is None:
                dst.zero_()
            elif dense and g.stride(1) == 1 and g.is_cuda:
                check(lib.hn_copy(_ptr(g), g.stride(0), ctypes.c_void_p(full.data_ptr() + 4 * a), ctot, cout, f, _stream()), "hn_copy")
            else:
                dst.copy_(g)
            a += f
        return (full,) + (None,) * len(ctx.sizes)


def split_columns(w, sizes):
    """Consecutive input-channel slices of a [Cout, sum(sizes), 1, 1] weight with a single fused gradient (see _SplitColumns)."""
    return _SplitColumns.apply(w, *[int(f) for f in sizes])


def concat(xs):
    xs = [as_f32(x) for x in xs]
    out = _Concat.apply(*xs)
    hits = [getattr(x, "_hn_scale", None) for x in xs]
    if all(h is not None and h[0] == x._version for h, x in zip(hits, xs)):
        # slot words are bit patterns of non-negative floats: their elementwise maximum bounds the concatenation
        m = hits[0][1]
        for h in hits[1:]:
            m = torch.maximum(m, h[1])
        out._hn_scale = (out._version, m)
    return out


# ------------------------------------------------------------------------------------------------------
# operators of the ResNeXt / ASPP "custom" architecture (cm/models/trgb_segnet.py)
# ------------------------------------------------------------------------------------------------------

class _GroupedConv3x3(torch.autograd.Function):
    """nn.Conv2d(C, C, 3, padding=dil, dilation=dil, groups=G, bias=False) with 2..16 channels per group
    (cm/models/trgb_segnet.py:106-107), direct fp32 kernels (csrc/grouped_conv.hip)."""

    @staticmethod
    def forward(ctx, x, w, groups, dil):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        w_in, w = w, weight_ohwi(w)
        n, c, h, wd = x.shape
        y = new_nhwc(n, c, h, wd, x.device)
        check(lib.hn_grouped_conv3x3_fwd(_ptr(x), _ptr(w), _ptr(y), n, h, wd, c, groups, ldx, c, dil, _stream()),
              "hn_grouped_conv3x3_fwd")
        ctx.save_for_backward(x, w)
        ctx.cfg = (groups, dil)
        ctx.w_leaf = _leaf(w_in)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        groups, dil = ctx.cfg
        x, ldx = as_nhwc(x)
        dy, lddy = as_nhwc(dy)
        n, c, h, wd = x.shape
        s = _stream()
        dx = dw_out = None
        if ctx.needs_input_grad[0]:
            dx = new_nhwc(n, c, h, wd, x.device)
            check(lib.hn_grouped_conv3x3_dgrad(_ptr(dy), _ptr(w), _ptr(dx), n, h, wd, c, groups, lddy, c, dil, s),
                  "hn_grouped_conv3x3_dgrad")
        if ctx.needs_input_grad[1]:
            slot = grad_slot(ctx.w_leaf) if ctx.w_leaf is not None else None
            flags = 0
            if slot is not None and weight_ohwi(slot[0]) is slot[0]:
                dw = slot[0]
                flags = 1 if slot[1] else 2
                dw_out = dw if slot[1] else None
            else:
                dw_out = dw = torch.empty((c, 3, 3, c // groups), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
            check(lib.hn_grouped_conv3x3_wgrad(_ptr(x), _ptr(dy), _ptr(dw), n, h, wd, c, groups, ldx, lddy, dil, flags, s),
                  "hn_grouped_conv3x3_wgrad")
        return dx, dw_out, None, None


def grouped_conv3x3(x, w, groups, dil=1):
    return _GroupedConv3x3.apply(x, w, int(groups), int(dil))


class _ConvTranspose2d(torch.autograd.Function):
    """nn.ConvTranspose2d(Cin, Cout, K, stride, pad, bias=False) (cm/models/trgb_segnet.py:208,338: the decoder's learned
    upsamplings, 13 channels): y = the DATA GRADIENT of the convolution whose OIHW weight is the transposed convolution's
    (Cin, Cout, K, K) tensor, so the forward runs hn_conv2d_dgrad, the input gradient hn_conv2d_fwd and the weight gradient
    hn_conv2d_wgrad with the roles of the two tensors swapped -- all on the generic fp32 kernels (13 channels)."""

    @staticmethod
    def forward(ctx, x, w, stride, pad):
        lib = _lib.load()
        x, ldx = as_nhwc(x)
        w_in, w = w, weight_ohwi(w)
        n, cin, h, wd = x.shape
        cin_w, cout, kh, kw = w.shape  # as a convolution: Cout_conv = cin, Cin_conv = cout
        ho, wo = (h - 1) * stride - 2 * pad + kh, (wd - 1) * stride - 2 * pad + kw
        y = new_nhwc(n, cout, ho, wo, x.device)
        s = _stream()
        wt = torch.empty(w.numel(), device=x.device, dtype=torch.float32)
        check(lib.hn_weight_flip_transpose(_ptr(w), None, _ptr(wt), None, cin, kh, kw, cout, stride, pad, 1, s),
              "hn_weight_flip_transpose")
        d = _desc(n, ho, wo, cout, cout, h, wd, cin, ldx, kh, kw, stride, pad, 1)  # the convolution y -> x
        check(lib.hn_conv2d_dgrad(_ptr(x), _ptr(wt), None, _ptr(y), None, 0, ctypes.byref(d), _variant(0, 0, 0, 1), s),
              "hn_conv2d_dgrad")
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, ho, wo)
        ctx.w_leaf = _leaf(w_in)
        return y

    @staticmethod
    def backward(ctx, gy):
        lib = _lib.load()
        x, w = ctx.saved_tensors
        stride, pad, ho, wo = ctx.cfg
        x, ldx = as_nhwc(x)
        gy, ldg = as_nhwc(gy)
        n, cin, h, wd = x.shape
        _, cout, kh, kw = w.shape
        s = _stream()
        f32 = _variant(0, 0, 0, 1)
        dx = dw_out = None
        if ctx.needs_input_grad[0]:  # dx = conv(gy, w) with the forward stride
            dx = new_nhwc(n, cin, h, wd, x.device)
            d = _desc(n, ho, wo, cout, ldg, h, wd, cin, cin, kh, kw, stride, pad, 1)
            check(lib.hn_conv2d_fwd(_ptr(gy), _ptr(w), None, None, _ptr(dx), None, ctypes.byref(d), ACT_NONE, 0.0, f32, s),
                  "hn_conv2d_fwd")
        if ctx.needs_input_grad[1]:  # dw[cin][k][k][cout] = wgrad(input = gy, output gradient = x)
            slot = grad_slot(ctx.w_leaf) if ctx.w_leaf is not None else None
            flags = 0
            if slot is not None and weight_ohwi(slot[0]) is slot[0]:
                dw = slot[0]
                flags = 1 if slot[1] else 2
                dw_out = dw if slot[1] else None
            else:
                dw_out = dw = torch.empty((cin, kh, kw, cout), device=x.device, dtype=torch.float32).permute(0, 3, 1, 2)
            d = _desc(n, ho, wo, cout, ldg, h, wd, cin, ldx, kh, kw, stride, pad, 1)
            check(lib.hn_conv2d_wgrad(_ptr(gy), _ptr(x), _ptr(dw), ctypes.byref(d), 0, f32, flags, s), "hn_conv2d_wgrad")
        return dx, dw_out, None, None


def conv_transpose2d(x, w, stride, pad):
    return _ConvTranspose2d.apply(x, w, int(stride), int(pad))


class _Add(torch.autograd.Function):
    """a + b (the residual sum of the pre-activation blocks, cm/models/trgb_segnet.py:127): hn_add; the gradient passes to both."""

    @staticmethod
    def forward(ctx, a, b):
        a, lda = as_nhwc(a)
        b, ldb = as_nhwc(b)
        n, c, h, w = a.shape
        y = new_nhwc(n, c, h, w, a.device)
        check(_lib.load().hn_add(_ptr(a), lda, _ptr(b), ldb, _ptr(y), c, n * h * w, c, _stream()), "hn_add")
        return y

    @staticmethod
    def backward(ctx, g):
        return g, g


def add(a, b):
    return _Add.apply(as_f32(a), as_f32(b))


# ------------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------------

class _CrossEntropy(torch.autograd.Function):
    """nn.CrossEntropyLoss(ignore_index=...) with mean reduction over the counted pixels."""

    @staticmethod
    def forward(ctx, logits, labels, ignore_index):
        lib = _lib.load()
        logits, ld = as_nhwc(logits)
        n, c, h, w = logits.shape
        if labels.dtype != torch.int64 or not labels.is_cuda:
            raise RuntimeError("cross_entropy: labels must be a cuda int64 tensor")
        labels = labels.contiguous()
        if labels.numel() != n * h * w:
            raise RuntimeError("cross_entropy: labels do not match logits")
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        check(lib.hn_ce_fwd(_ptr(logits), ld, _ptr(labels), n * h * w, c, ignore_index, _ptr(out), _stream()),
              "hn_ce_fwd")
        ctx.save_for_backward(logits, labels, out)
        ctx.ignore = ignore_index
        return out[0] / out[1]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logits, labels, out = ctx.saved_tensors
        logits, ld = as_nhwc(logits)
        n, c, h, w = logits.shape
        g = g.contiguous().reshape(1)
        dl = new_nhwc(n, c, h, w, logits.device)
        check(lib.hn_ce_bwd(_ptr(logits), ld, _ptr(labels), n * h * w, c, ctx.ignore, _ptr(out), _ptr(g), _ptr(dl), c,
                            _stream()), "hn_ce_bwd")
        return dl, None, None


def cross_entropy(logits, labels, ignore_index=-100):
    return _CrossEntropy.apply(logits, labels, int(ignore_index))


class _CrossEntropyWeighted(torch.autograd.Function):
    """torch.mean(weight * nn.CrossEntropyLoss(reduction='none')(logits, labels)) with a constant per-pixel weight map
    (`--weight_ir_sup`, cm/train_trgb_segnet_conf.py:468-477)."""

    @staticmethod
    def forward(ctx, logits, labels, weight):
        lib = _lib.load()
        logits, ld = as_nhwc(logits)
        n, c, h, w = logits.shape
        if labels.dtype != torch.int64 or not labels.is_cuda or labels.numel() != n * h * w:
            raise RuntimeError("cross_entropy_weighted: labels must be a cuda int64 tensor matching the logits")
        _require_gpu(weight)
        if weight.numel() != n * h * w:
            raise RuntimeError("cross_entropy_weighted: one weight per pixel")
        labels, weight = labels.contiguous(), weight.contiguous()
        out = torch.empty(2, device=logits.device, dtype=torch.float32)
        check(lib.hn_ce_weighted_fwd(_ptr(logits), ld, _ptr(labels), _ptr(weight), n * h * w, c, _ptr(out), _stream()),
              "hn_ce_weighted_fwd")
        ctx.save_for_backward(logits, labels, weight, out)
        return out[0] / out[1]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        logits, labels, weight, out = ctx.saved_tensors
        logits, ld = as_nhwc(logits)
        n, c, h, w = logits.shape
        g = g.contiguous().reshape(1)
        dl = new_nhwc(n, c, h, w, logits.device)
        check(lib.hn_ce_weighted_bwd(_ptr(logits), ld, _ptr(labels), _ptr(weight), n * h * w, c, _ptr(out), _ptr(g), _ptr(dl),
                                     c, _stream()), "hn_ce_weighted_bwd")
        return dl, None, None


def cross_entropy_weighted(logits, labels, weight):
    return _CrossEntropyWeighted.apply(logits, labels, weight)


class _Sigmoid(torch.autograd.Function):
    """torch.sigmoid on the certainty map of the custom architecture (cm/models/trgb_segnet.py:387)."""

    @staticmethod
    def forward(ctx, x):
        _require_gpu(x)
        x = x.contiguous()
        y = torch.empty_like(x)
        check(_lib.load().hn_sigmoid_fwd(_ptr(x), _ptr(y), x.numel(), _stream()), "hn_sigmoid_fwd")
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        check(_lib.load().hn_sigmoid_bwd(_ptr(y), _ptr(dy), _ptr(dx), y.numel(), _stream()), "hn_sigmoid_bwd")
        return dx


def sigmoid(x):
    return _Sigmoid.apply(as_f32(x))


class _CertLoss(torch.autograd.Function):
    """The trainer's certainty loss as written (cm/train_trgb_segnet_conf.py:482-491): 10 * mean((cert_a - (1 - p_label)) ** 2)
    with cert_a [N, 1, H, W] broadcast against the [N, H, W] target to [N, N, H, W]; the target is not detached, so the
    segmentation logits receive a gradient through it (hn_cert_loss_fwd / _bwd)."""

    @staticmethod
    def forward(ctx, logits, labels, cert):
        lib = _lib.load()
        logits, ld = as_nhwc(logits)
        n, c, h, w = logits.shape
        if tuple(cert.shape) != (n, 1, h, w) or tuple(labels.shape) != (n, h, w):
            raise RuntimeError("cert_loss: logits [N, C, H, W], labels [N, H, W], cert [N, 1, H, W]")
        _require_gpu(cert)
        cert = cert.contiguous()
        labels = labels.contiguous()
        out = torch.zeros(1, device=logits.device, dtype=torch.float32)
        check(lib.hn_cert_loss_fwd(_ptr(logits), ld, _ptr(labels), _ptr(cert), n, h * w, c, _ptr(out), _stream()), "hn_cert_loss_fwd")
        ctx.save_for_backward(logits, labels, cert)
        ctx.ld = ld
        return out[0] * (10.0 / (n * n * h * w))

    @staticmethod
    def backward(ctx, g):
        logits, labels, cert = ctx.saved_tensors
        n, c, h, w = logits.shape
        g = g.reshape(1).to(torch.float32).contiguous()
        dl = new_nhwc(n, c, h, w, logits.device)
        dc = torch.empty_like(cert) if ctx.needs_input_grad[2] else None
        check(_lib.load().hn_cert_loss_bwd(_ptr(logits), ctx.ld, _ptr(labels), _ptr(cert), n, h * w, c, _ptr(g), _ptr(dc), _ptr(dl), c,
                                           _stream()), "hn_cert_loss_bwd")
        return dl, None, dc


def cert_loss(logits, labels, cert):
    return _CertLoss.apply(logits, labels, cert)


def teacher_outputs(logits, want_cert=True):
    """(label int64 [N,H,W], cert fp32 [N,H,W] or None) of the frozen night-supervision teacher: arg-max of its softmax and
    the reference's certainty map max(softmax(softmax(logits))) (cm/train_trgb_segnet_conf.py:430-434, 468-472)."""
    lib = _lib.load()
    logits, ld = as_nhwc(logits)
    n, c, h, w = logits.shape
    label = torch.empty((n, h, w), device=logits.device, dtype=torch.int64)
    cert = torch.empty((n, h, w), device=logits.device, dtype=torch.float32) if want_cert else None
    check(lib.hn_teacher_cert(_ptr(logits), ld, n * h * w, c, _ptr(label), _ptr(cert), _stream()), "hn_teacher_cert")
    return label, cert


class _AdvLoss(torch.autograd.Function):
    """mean over all elements of MSE / BCE-with-logits against a constant target map."""

    @staticmethod
    def forward(ctx, x, target, kind):
        lib = _lib.load()
        _require_gpu(x)
        x = x.contiguous()
        out = torch.empty(1, device=x.device, dtype=torch.float32)
        check(lib.hn_adv_loss_fwd(_ptr(x), x.numel(), target, kind, _ptr(out), _stream()), "hn_adv_loss_fwd")
        ctx.save_for_backward(x)
        ctx.cfg = (target, kind)
        return out[0]

    @staticmethod
    def backward(ctx, g):
        lib = _lib.load()
        (x,) = ctx.saved_tensors
        target, kind = ctx.cfg
        g = g.contiguous().reshape(1)
        dx = torch.empty_like(x)
        check(lib.hn_adv_loss_bwd(_ptr(x), x.numel(), target, kind, _ptr(g), _ptr(dx), _stream()), "hn_adv_loss_bwd")
        return dx, None, None


def adv_loss(x, target, kind):
    if isinstance(kind, str):
        kind = {"MSE": LOSS_MSE, "BCE": LOSS_BCE}[kind]
    return _AdvLoss.apply(x, float(target), int(kind))


# ------------------------------------------------------------------------------------------------------
# evaluation / optimizer helpers (no autograd)
# ------------------------------------------------------------------------------------------------------

def argmax_confusion(logits, labels, num_classes):
    """(pred uint8 [N,H,W], conf int64 [K,K]) with conf[label, pred] counts; labels outside [0,K) are skipped."""
    lib = _lib.load()
    logits, ld = as_nhwc(logits)
    n, c, h, w = logits.shape
    pred = torch.empty((n, h, w), device=logits.device, dtype=torch.uint8)
    conf = torch.zeros((num_classes, num_classes), device=logits.device, dtype=torch.int64)
    lab = labels.contiguous() if labels is not None else None
    check(lib.hn_argmax_confusion(_ptr(logits), ld, _ptr(lab), n * h * w, c, num_classes, _ptr(conf), _ptr(pred),
                                  _stream()), "hn_argmax_confusion")
    return pred, conf


def rmsprop_step_(param, grad, square_avg, lr, alpha=0.99, eps=1e-8, weight_decay=0.0):
    lib = _lib.load()
    check(lib.hn_rmsprop_step(_ptr(param), _ptr(grad), _ptr(square_avg), param.numel(), lr, alpha, eps, weight_decay,
                              _stream()), "hn_rmsprop_step")


def adam_step_(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    lib = _lib.load()
    check(lib.hn_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr, beta1, beta2,
                           eps, weight_decay, step, _stream()), "hn_adam_step")


if os.environ.get("HEATNET_DETERMINISTIC", "0") == "1":
    DETERMINISTIC = True  # the library flag is raised on first use (importing must not require the library on a CPU host)
