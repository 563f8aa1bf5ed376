"""ctypes binding of libslamhip.so (C ABI: include/slamhip.h).

Thin by design: torch tensors own all device memory; this module only passes
`data_ptr()`s, sizes and the current HIP stream, and turns non-zero return
codes into RuntimeError.  A missing library is a hard error (no fallback).
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SLAMHIP_LIB") or os.path.join(_HERE, "_lib", "libslamhip.so")

NSUMS = 4
NCOUNTS = 4
STATIC_MAX_PAIRS = 8192
M_VALID0, M_INB, M_WARP, M_GEO = 1, 2, 4, 8

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_double = ctypes.c_double
_K4 = ctypes.c_double * 4

# name -> argtypes, exactly the declarations of include/slamhip.h
SIGNATURES = {
    "slamhip_last_error": ([], ctypes.c_char_p),
    "slamhip_version": ([], _c_int),
    "slamhip_deterministic": ([], _c_int),
    "slamhip_zero_fill": ([_c_void_p, ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_caller_clears_accumulators": ([_c_int], _c_int),
    "slamhip_quat_to_transformation": ([_c_void_p, _c_int, _c_void_p], _c_int),
    "slamhip_inverse_trans": ([_c_void_p, _c_int, _c_void_p], _c_int),
    "slamhip_absolute_position": ([_c_void_p, _c_int, _c_void_p], _c_int),
    "slamhip_trans_to_twist": ([_c_void_p, _c_int, _c_int, _c_void_p], _c_int),
    "slamhip_twist_to_trans": ([_c_void_p, _c_int, _c_int, _c_void_p], _c_int),
    "slamhip_pose_batch": ([_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p, _c_void_p], _c_int),
    "slamhip_relative_pose_device": ([_c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p], _c_int),
    "slamhip_backproject": ([_c_void_p, _c_int, _c_int, _c_int, _K4, _c_double, _c_void_p, _c_int,
                             _c_void_p, _c_void_p], _c_int),
    "slamhip_reproject": ([_c_void_p, _c_int, _c_void_p, _c_void_p, _K4, _c_double, _c_double, _c_int,
                           _c_int, _c_int, _c_void_p, _c_void_p, _c_int, _c_void_p, _c_void_p, _c_void_p,
                           _c_void_p], _c_int),
    "slamhip_overlap_count": ([_c_void_p, ctypes.c_longlong, _c_double, _c_double, _c_void_p, _c_void_p,
                               _c_void_p], _c_int),
    "slamhip_overlap_fused": ([_c_void_p, _c_void_p, _K4, _c_double, _c_double, _c_double, _c_int, _c_int, _c_int, _c_void_p,
                               _c_void_p], _c_int),
    "slamhip_overlap_fused_workspace_bytes": ([_c_int], ctypes.c_size_t),
    "slamhip_overlap_fused_ws": ([_c_void_p, _c_void_p, _K4, _c_double, _c_double, _c_double, _c_int, _c_int, _c_int,
                                  _c_void_p, _c_void_p, ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_pair_loss_workspace_bytes": ([_c_int, _c_int], ctypes.c_size_t),
    "slamhip_pair_loss_fused": ([_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _K4, _c_double,
                                 _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                 ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_pair_loss_fused_static": ([_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _K4, _c_double,
                                        _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p], _c_int),
    "slamhip_pair_loss_reset_workspace": ([_c_void_p, ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_warp_materialize": ([_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _K4, _c_double,
                                  _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                  _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p], _c_int),
    "slamhip_conv3x3_igemm_bf16": ([_c_void_p] * 4 + [_c_int] * 6 + [_c_void_p], _c_int),
    "slamhip_conv_stats_workspace_bytes": ([ctypes.c_longlong, _c_int], ctypes.c_size_t),
    "slamhip_conv3x3_igemm_bf16_stats": ([_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p, _c_void_p, ctypes.c_size_t,
                                                                        _c_void_p], _c_int),
    "slamhip_conv3x3_small_fwd_stats": ([_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p, _c_void_p, ctypes.c_size_t,
                                                                       _c_void_p], _c_int),
    "slamhip_gemm_nt_bf16": ([_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p], _c_int),
    "slamhip_wgrad_bf16": ([_c_void_p] * 3 + [_c_int] * 6 + [_c_void_p], _c_int),
    "slamhip_conv3x3_small_fwd": ([_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p], _c_int),
    "slamhip_conv3x3_small_wgrad": ([_c_void_p] * 3 + [_c_int] * 5 + [_c_void_p], _c_int),
    "slamhip_bn_relu_bwd_first_layer": ([_c_void_p] * 8 + [_c_int] * 3 + [ctypes.c_float, _c_void_p], _c_int),
    "slamhip_conv3x3_c4_bn_relu_fwd": ([_c_void_p] * 7 + [ctypes.c_size_t] + [_c_int] * 3 + [ctypes.c_float, _c_void_p], _c_int),
    "slamhip_bn_relu_bwd_first_layer_recompute": ([_c_void_p] * 9 + [ctypes.c_size_t] + [_c_int] * 3 +
                                                  [ctypes.c_float, _c_void_p], _c_int),
    "slamhip_channel_stats": ([_c_void_p, _c_void_p, ctypes.c_longlong, _c_int, _c_void_p], _c_int),
    "slamhip_bn_relu_fwd": ([_c_void_p] * 5 + [ctypes.c_longlong, _c_int, ctypes.c_float, _c_void_p], _c_int),
    "slamhip_bn_relu_bwd": ([_c_void_p] * 7 + [ctypes.c_longlong, _c_int, ctypes.c_float, _c_void_p], _c_int),
    "slamhip_bn_relu_pool_fwd": ([_c_void_p] * 6 + [_c_int] * 4 + [ctypes.c_float, _c_void_p], _c_int),
    "slamhip_bn_relu_pool_bwd": ([_c_void_p] * 7 + [_c_int] * 4 + [ctypes.c_float, _c_void_p], _c_int),
    "slamhip_bn_relu_bwd_phases": ([_c_void_p] * 7 + [ctypes.c_longlong, _c_int, ctypes.c_float, _c_int, _c_void_p], _c_int),
    "slamhip_bn_relu_pool_bwd_phases": ([_c_void_p] * 7 + [_c_int] * 4 + [ctypes.c_float, _c_int, _c_void_p], _c_int),
    "slamhip_bias_act_fwd": ([_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p], _c_int),
    "slamhip_bias_act_bwd": ([_c_void_p] * 4 + [_c_int] * 3 + [_c_void_p], _c_int),
    "slamhip_maxpool2_fwd": ([_c_void_p] * 2 + [_c_int] * 4 + [_c_void_p], _c_int),
    "slamhip_maxpool2_bwd": ([_c_void_p] * 4 + [_c_int] * 4 + [_c_void_p], _c_int),
    "slamhip_cast_f32_bf16": ([_c_void_p, _c_void_p, ctypes.c_longlong, _c_void_p], _c_int),
    "slamhip_transpose_cast_bf16": ([_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p], _c_int),
    "slamhip_transpose_bf16": ([_c_void_p, _c_void_p, _c_int, _c_int, _c_void_p], _c_int),
    "slamhip_conv_weight_layouts": ([_c_void_p] * 3 + [_c_int] * 2 + [_c_void_p], _c_int),
    "slamhip_conv_weight_layouts_multi": ([_c_void_p, _c_int] + [_c_void_p] * 6, _c_int),
    "slamhip_l2loss": ([_c_void_p] * 5 + [_c_int] * 2 + [_c_void_p], _c_int),
    "slamhip_weighted_sq_loss": ([_c_void_p] * 3 + [ctypes.c_float] + [_c_void_p] * 2 + [_c_int] * 2 + [_c_void_p], _c_int),
    "slamhip_posenet_loss": ([_c_void_p] * 2 + [ctypes.c_float] + [_c_void_p] * 2 + [_c_int, _c_void_p], _c_int),
    "slamhip_bias_relu_fwd": ([_c_void_p] * 3 + [ctypes.c_longlong, _c_int, _c_void_p], _c_int),
    "slamhip_bias_relu_bwd": ([_c_void_p] * 4 + [ctypes.c_longlong, _c_int, _c_void_p], _c_int),
    "slamhip_prep_rgbd_workspace_bytes": ([_c_int], ctypes.c_size_t),
    "slamhip_prep_rgbd": ([_c_void_p, _c_void_p] + [_c_int] * 5 + [_c_void_p] * 4 + [ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_prep_rgbd_tf": ([_c_void_p, _c_void_p] + [_c_int] * 5 + [_c_void_p] * 2 + [ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_sgd_step": ([_c_void_p, _c_void_p, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, _c_void_p], _c_int),
    "slamhip_rmsprop_step": ([_c_void_p] * 3 + [ctypes.c_longlong] + [ctypes.c_float] * 4 + [_c_void_p], _c_int),
    "slamhip_lstm_head_fwd": ([_c_void_p] * 6 + [ctypes.c_float] + [_c_int] * 4 + [_c_void_p] * 4, _c_int),
    "slamhip_lstm_head_bwd": ([_c_void_p] * 7 + [_c_int] * 4 + [_c_void_p] * 7, _c_int),
    "slamhip_conv_out_shape": ([_c_int] * 6 + [_c_void_p] * 2, _c_int),
    "slamhip_im2col_bf16": ([_c_void_p] * 2 + [_c_int] * 7 + [_c_void_p], _c_int),
    "slamhip_col2im_add_bf16": ([_c_void_p] * 2 + [_c_int] * 7 + [_c_void_p], _c_int),
    "slamhip_bias_act_slice_fwd": ([_c_void_p] * 4 + [ctypes.c_longlong] + [_c_int] * 5 + [_c_void_p], _c_int),
    "slamhip_bias_act_slice_bwd": ([_c_void_p] * 5 + [ctypes.c_longlong] + [_c_int] * 5 + [_c_void_p], _c_int),
    "slamhip_pool_fwd": ([_c_void_p] * 2 + [_c_int] * 8 + [_c_void_p], _c_int),
    "slamhip_pool_bwd_add": ([_c_void_p] * 3 + [_c_int] * 8 + [_c_void_p], _c_int),
    "slamhip_lrn_fwd": ([_c_void_p] * 2 + [ctypes.c_longlong, _c_int, _c_int] + [ctypes.c_float] * 3 + [_c_void_p], _c_int),
    "slamhip_lrn_bwd_add": ([_c_void_p] * 3 + [ctypes.c_longlong, _c_int, _c_int] + [ctypes.c_float] * 3 + [_c_void_p], _c_int),
    "slamhip_pad_cast_bf16": ([_c_void_p] * 2 + [ctypes.c_longlong, _c_int, _c_int, _c_void_p], _c_int),
    "slamhip_adam_step": ([_c_void_p] * 4 + [ctypes.c_longlong] + [ctypes.c_float] * 4 + [_c_int, ctypes.c_float,
                                                                                          _c_void_p], _c_int),
    "slamhip_adam_step_dev": ([_c_void_p] * 4 + [ctypes.c_longlong, _c_void_p] + [ctypes.c_float] * 4 +
                              [_c_void_p, ctypes.c_longlong, _c_void_p], _c_int),
    "slamhip_adam_step_shadow": ([_c_void_p] * 4 + [ctypes.c_longlong] + [ctypes.c_float] * 4 +
                                 [_c_int, ctypes.c_float, _c_void_p, ctypes.c_longlong, _c_void_p], _c_int),
}

_lib = None


def load():
    """Load libslamhip.so once and type its entry points."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "slam_amd: %s is missing - build it with `make -C slam_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().slamhip_last_error()
        raise RuntimeError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))


def K4(K):
    return _K4(*[float(x) for x in K])


def ptr(t):
    """Device/host address of a torch tensor or numpy array (None -> NULL)."""
    if t is None:
        return None
    if hasattr(t, "data_ptr"):
        return t.data_ptr()
    return t.ctypes.data


def current_stream():
    import torch
    return torch.cuda.current_stream().cuda_stream
