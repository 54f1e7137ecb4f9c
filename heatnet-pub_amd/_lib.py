"""ctypes binding of libheatnet_hip.so (declared in include/heatnet_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C heatnet-pub_amd/csrc`.
There is NO fallback: if the shared object is missing or a symbol is absent, importing
the operators raises, and every operator refuses CPU tensors.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# HEATNET_HIP_LIB selects another build of the same library (kernel experiments); it is never a fallback
LIB_PATH = os.environ.get("HEATNET_HIP_LIB") or os.path.join(_HERE, "lib", "libheatnet_hip.so")

c_int, c_i64, c_float, c_void_p = ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p
P = c_void_p  # every device pointer crosses the ABI as a plain address


class HnConvDesc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("N", "Hin", "Win", "Cin", "ldx", "Hout", "Wout", "Cout", "ldy", "KH", "KW", "stride", "pad", "dil")]


DP = ctypes.POINTER(HnConvDesc)


HN_ESHAPE = -4  # include/heatnet_hip.h


class HnBnReduceArgs(ctypes.Structure):
    """include/heatnet_hip.h: the BatchNorm-backward reduction that rides in a data-gradient epilogue."""
    _fields_ = [("y", ctypes.c_void_p), ("ldy", ctypes.c_int32), ("y_bf16", ctypes.c_int32), ("act", ctypes.c_int32),
                ("reserved", ctypes.c_int32), ("scale", ctypes.c_void_p), ("shift", ctypes.c_void_p), ("mean", ctypes.c_void_p),
                ("invstd", ctypes.c_void_p), ("relu_bits", ctypes.c_void_p), ("sums", ctypes.c_void_p), ("du_max", ctypes.c_void_p)]


class HnWeightPrepItem(ctypes.Structure):
    _fields_ = [("w", ctypes.c_void_p), ("w_x3", ctypes.c_void_p), ("wt", ctypes.c_void_p), ("wt_x3", ctypes.c_void_p)] + \
               [(n, ctypes.c_int32) for n in ("Cout", "KH", "KW", "Cin", "stride", "pad", "dil", "format")]


# name -> argtypes, exactly the prototypes of include/heatnet_hip.h
SIGNATURES = {
    "hn_conv2d_fwd": [P, P, P, P, P, P, DP, c_int, c_float, c_int, P],
    "hn_weight_flip_transpose": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_conv2d_dgrad": [P, P, P, P, P, c_int, DP, c_int, P],
    "hn_conv2d_wgrad": [P, P, P, DP, c_int, c_int, c_int, P],
    "hn_conv2d_wgrad_det": [P, P, P, DP, c_int, c_int, c_int, P, P],
    "hn_channel_sum": [P, c_int, c_i64, c_int, P, P],
    "hn_channel_stats": [P, c_int, c_i64, c_int, P, P],
    "hn_bn_finalize": [P, c_i64, P, P, P, P, c_float, c_float, P, P, P, P, c_int, P],
    "hn_bn_eval_params": [P, P, P, P, c_float, P, P, P, P, c_int, P],
    "hn_bn_act_fwd": [P, c_int, P, P, P, c_int, P, c_int, c_i64, c_int, c_int, P, c_float, P, c_i64, P, P, P],
    "hn_bn_act_bwd_reduce": [P, c_int, P, c_int, P, c_int, P, P, P, P, c_i64, c_int, c_int, P, c_float, P, P, P, P, P,
                             c_int, P, c_i64, P, P],
    "hn_bn_act_bwd_apply": [P, c_int, P, c_int, P, c_int, P, P, P, P, P, P, c_i64, c_int, P, c_int, P, c_int,
                            c_i64, c_int, c_int, P, c_float, P, c_i64, P, P, P],
    "hn_act_bwd": [P, c_int, P, c_int, P, c_int, c_i64, c_int, c_int, c_float, P],
    "hn_act_bwd_bias": [P, c_int, P, c_int, P, c_int, c_i64, c_int, c_int, c_float, P, P, P],
    "hn_scale_nc": [P, c_int, P, P, c_int, c_int, c_i64, c_int, P],
    "hn_add": [P, c_int, P, c_int, P, c_int, c_i64, c_int, P],
    "hn_dropout_mask": [P, c_int, c_float, ctypes.c_uint64, P],
    "hn_nchw_to_nhwc": [P, P, c_int, c_int, c_i64, c_int, P],
    "hn_copy": [P, c_int, P, c_int, c_i64, c_int, P],
    "hn_maxpool3x3s2_fwd": [P, c_int, P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_maxpool3x3s2_bwd": [P, c_int, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_adaptive_avgpool_fwd": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_adaptive_avgpool_bwd": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_bilinear_fwd": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_bilinear_bwd": [P, c_int, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_pyramid_merge_fwd": [P, c_int, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_split_bf16x3": [P, P, c_i64, P],
    "hn_split_f16x2": [P, P, c_i64, P],
    "hn_absmax_scale": [P, c_int, c_i64, c_int, P, P],
    "hn_conv2d_fwd_scaled": [P, P, P, P, P, P, P, DP, c_int, c_float, c_int, P],
    "hn_conv2d_dgrad_scaled": [P, P, P, P, P, P, c_int, DP, c_int, P],
    "hn_weight_prep_table_fill": [P, c_int, P, P],
    "hn_weight_prepare_batch": [P, c_int, c_int, c_int, P],
    "hn_weight_flip_transpose_h2": [P, P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_conv2d_wgrad_scaled": [P, P, P, P, P, DP, c_int, c_int, c_int, P],
    "hn_conv2d_wgrad_det_scaled": [P, P, P, P, P, DP, c_int, c_int, c_int, P, P],
    "hn_tensor_to_planes": [P, c_int, c_i64, c_int, P, P, c_i64, c_int, P],
    "hn_planes_to_tensor": [P, c_i64, c_int, P, P, c_int, c_i64, c_int, P],
    "hn_conv2d_fwd_planes": [P, c_i64, P, P, P, P, P, P, DP, c_int, c_float, c_int, P],
    "hn_conv2d_fwd_scaled_range": [P, P, P, P, P, P, P, P, DP, c_int, P],
    "hn_conv2d_dgrad_planes": [P, c_i64, P, P, P, P, c_int, DP, c_int, P],
    "hn_conv2d_dgrad_planes_bnreduce": [P, c_i64, P, P, P, P, c_int, DP, c_int, ctypes.POINTER(HnBnReduceArgs), P],
    "hn_bn_act_bwd_finish": [P, P, c_int, P, P, P, c_int, P, P, P, P, c_i64, c_int, P, P],
    "hn_conv2d_wgrad_planes": [P, c_i64, P, P, c_i64, P, P, DP, c_int, c_int, c_int, P],
    "hn_conv2d_wgrad_det_planes": [P, c_i64, P, P, c_i64, P, P, DP, c_int, c_int, c_int, P, P],
    "hn_bn_finalize_range": [P, P, c_i64, P, P, P, P, c_float, c_float, P, P, P, P, P, c_int, c_int, P, c_float, P, c_float, P, P],
    "hn_bn_act_fwd_planes": [P, c_int, P, P, P, c_int, P, c_i64, c_int, P, P, c_int, P, c_i64, c_int, P, c_i64, c_int, c_int, P,
                             c_float, P, c_i64, P, P],
    "hn_bn_act_bwd_reduce_range": [P, c_int, P, c_int, P, c_int, P, P, P, P, c_i64, c_int, c_int, P, c_float, P, P, P, P, P,
                                   c_int, P, c_i64, P, P, P, P, c_i64, c_int, P, P],
    "hn_bn_act_bwd_apply_planes": [P, c_int, P, c_int, P, c_int, P, P, P, P, P, P, c_i64, c_int, P, c_i64, c_int, P, P, c_int,
                                   c_i64, c_int, c_int, P, c_float, P, c_i64, P, P],
    "hn_bilinear_up2_planes": [P, c_int, P, P, c_i64, c_int, c_int, c_int, c_int, c_int, P],
    "hn_range_census": [P, P, c_i64, c_int, P, c_i64, c_int, P, P],
    "hn_conv_stem_fwd": [P, P, P, P, DP, P],
    "hn_conv_stem_wgrad": [P, P, P, DP, c_int, P],
    "hn_conv_narrow_dgrad": [P, P, P, DP, P],
    "hn_conv_narrow_wgrad": [P, P, P, DP, c_int, P],
    "hn_grouped_conv3x3_fwd": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_grouped_conv3x3_dgrad": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_grouped_conv3x3_wgrad": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P],
    "hn_thermal_augment": [P, P, P, c_int, c_int, c_int, P, P, c_int, c_int, P, P, P, P],
    "hn_thermal_resize": [P, P, P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, P, P, P, P],
    "hn_set_conv_math": [c_int],
    "hn_get_conv_math": [],
    "hn_set_deterministic": [c_int],
    "hn_get_deterministic": [],
    "hn_conv2d_fwd_affine": [P, P, P, P, P, P, P, P, DP, c_int, P],
    "hn_conv2d_wgrad_plan": [DP, c_int, c_int, ctypes.c_char_p, c_int],
    "hn_conv2d_plan": [c_int, DP, c_int, c_int, ctypes.c_char_p, c_int],
    "hn_pyramid_merge_bwd_segments": [c_int, c_int, c_int],
    "hn_pyramid_merge_bwd": [P, c_int, P, c_int, P, c_int, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "hn_pyramid_pool_fwd": [P, c_int, c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, P],
    "hn_pyramid_pool_bwd": [c_int, P, P, P, P, P, c_int, c_int, c_int, c_int, c_int, P],
    "hn_ce_fwd": [P, c_int, P, c_i64, c_int, c_i64, P, P],
    "hn_ce_bwd": [P, c_int, P, c_i64, c_int, c_i64, P, P, P, c_int, P],
    "hn_ce_weighted_fwd": [P, c_int, P, P, c_i64, c_int, P, P],
    "hn_ce_weighted_bwd": [P, c_int, P, P, c_i64, c_int, P, P, P, c_int, P],
    "hn_teacher_cert": [P, c_int, c_i64, c_int, P, P, P],
    "hn_adv_loss_fwd": [P, c_i64, c_float, c_int, P, P],
    "hn_adv_loss_bwd": [P, c_i64, c_float, c_int, P, P, P],
    "hn_sigmoid_fwd": [P, P, c_i64, P],
    "hn_sigmoid_bwd": [P, P, P, c_i64, P],
    "hn_cert_loss_fwd": [P, c_int, P, P, c_int, c_i64, c_int, P, P],
    "hn_cert_loss_bwd": [P, c_int, P, P, c_int, c_i64, c_int, P, P, P, c_int, P],
    "hn_argmax_confusion": [P, c_int, P, c_i64, c_int, c_int, P, P, P],
    "hn_rmsprop_step": [P, P, P, c_i64, c_float, c_float, c_float, c_float, P],
    "hn_adam_step": [P, P, P, P, c_i64, c_float, c_float, c_float, c_float, c_float, c_int, P],
}

_lib = None


def load():
    """Load the shared object once; raise (never fall back) when it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "heatnet_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C heatnet-pub_amd/csrc`). There is no CPU/PyTorch fallback for the hot path." % LIB_PATH)
    # PyTorch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import torch FIRST so that this library
    # binds to the HIP runtime torch's streams and allocations belong to; loaded the other way round the process ends
    # up with the system runtime under torch and kernel launches on torch's streams fail.
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = c_int
    lib.hn_strerror.argtypes = [c_int]
    lib.hn_strerror.restype = ctypes.c_char_p
    lib.hn_weight_prep_table_bytes.argtypes = [c_int]
    lib.hn_weight_prep_table_bytes.restype = ctypes.c_size_t
    lib.hn_version.argtypes = []
    lib.hn_version.restype = c_int
    if os.environ.get("HEATNET_DETERMINISTIC", "0") == "1":
        lib.hn_set_deterministic(1)  # ops.DETERMINISTIC mirrors this (no split-K float atomics)
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise RuntimeError("heatnet_amd: %s failed: %s (%d)" % (what, load().hn_strerror(code).decode(), code))
