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
M_VALID0, M_INB, M_WARP, M_GEO = 1, 2, 4, 8

_c_void_p = ctypes.c_void_p
_c_int = ctypes.c_int
_c_double = ctypes.c_double
_K4 = ctypes.c_double * 4

# name -> argtypes, exactly the declarations of include/slamhip.h
SIGNATURES = {
    "slamhip_last_error": ([], ctypes.c_char_p),
    "slamhip_version": ([], _c_int),
    "slamhip_quat_to_transformation": ([_c_void_p, _c_int, _c_void_p], _c_int),
    "slamhip_inverse_trans": ([_c_void_p, _c_int, _c_void_p], _c_int),
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
    "slamhip_pair_loss_workspace_bytes": ([_c_int, _c_int], ctypes.c_size_t),
    "slamhip_pair_loss_fused": ([_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _K4, _c_double,
                                 _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                 ctypes.c_size_t, _c_void_p], _c_int),
    "slamhip_warp_materialize": ([_c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p, _K4, _c_double,
                                  _c_int, _c_int, _c_int, _c_int, _c_void_p, _c_void_p, _c_void_p,
                                  _c_void_p, _c_void_p, _c_void_p, _c_void_p, _c_void_p], _c_int),
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
