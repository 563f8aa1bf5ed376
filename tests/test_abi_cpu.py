"""CPU-only checks of the boundary: libslamhip.so loads without a GPU, exports
every function include/slamhip.h declares, the ctypes table matches the header,
argument validation returns error codes, and the host (CPU) pose entry points
reproduce the reference's golden vectors.  No kernel is launched here.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from slam_amd import capi
from slam_amd import frame_pairs as fp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "slamhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)          # comments mention names too
    return sorted(set(re.findall(r"\b(slamhip_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    names = header_functions()
    assert "slamhip_pair_loss_fused" in names and "slamhip_conv3x3_igemm_bf16" in names
    assert len(names) >= 30


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(capi.LIB_PATH)
    missing = [n for n in header_functions() if not hasattr(lib, n)]
    assert not missing, "declared in include/slamhip.h but not exported: %s" % missing


def test_ctypes_table_matches_header():
    assert sorted(capi.SIGNATURES) == header_functions()


def test_argument_counts_match_header():
    text = open(os.path.join(ROOT, "include", "slamhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    for name, (argtypes, _) in capi.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, text, flags=re.S)
        assert m, name
        params = m.group(1).strip()
        n = 0 if params in ("", "void") else len(params.split(","))
        assert n == len(argtypes), "%s: header has %d parameters, ctypes table %d" % (name, n, len(argtypes))


def test_version_and_error_text():
    lib = capi.load()
    assert lib.slamhip_version() >= 1
    # a null pointer is an argument error with a message, not a crash
    rc = lib.slamhip_quat_to_transformation(None, 3, None)
    assert rc == -1
    assert b"quat" in lib.slamhip_last_error().lower() or lib.slamhip_last_error()
    # device entry points validate before touching the GPU
    rc = lib.slamhip_pair_loss_fused(None, None, None, None, None, capi.K4([1, 1, 0, 0]), 1.0, 2, 1, 480, 640,
                                     None, None, None, 0, None)
    assert rc == -1
    rc = lib.slamhip_conv3x3_igemm_bf16(None, None, None, None, 1, 8, 8, 64, 64, 0, None)
    assert rc == -1
    rc = lib.slamhip_wgrad_bf16(ctypes.c_void_p(8), ctypes.c_void_p(8), ctypes.c_void_p(8), 1, 4, 4, 48, 64, 9, None)
    assert rc == -3       # channels not a multiple of 64
    # the optimizer's shadow copy may not be longer than the parameters; null operands are errors
    p8 = ctypes.c_void_p(8)
    rc = lib.slamhip_adam_step_shadow(p8, p8, p8, p8, 10, 0.01, 0.9, 0.999, 1e-8, 1, 1.0, p8, 11, None)
    assert rc == -1 and b"shadow" in lib.slamhip_last_error()
    rc = lib.slamhip_adam_step_shadow(None, p8, p8, p8, 10, 0.01, 0.9, 0.999, 1e-8, 1, 1.0, None, 0, None)
    assert rc == -1
    assert lib.slamhip_transpose_bf16(None, p8, 8, 8, None) == -1
    # the launch grid's extents bound a call: (tile rows x neighbours) and groups of eight anchors at most 65535
    a16 = ctypes.c_void_p(4096)
    for B, k, H in ((1, 70000, 24), (8 * 65535 + 1, 1, 24), (1, 100, 24 * 700)):
        need = lib.slamhip_pair_loss_workspace_bytes(B, k)
        rc = lib.slamhip_pair_loss_fused(a16, a16, a16, a16, a16, capi.K4([500, 500, 64, 12]), 1.0, B, k, H, 128,
                                         a16, a16, a16, need, None)
        assert rc == -3 and b"grid too large" in lib.slamhip_last_error(), (B, k, H, rc, lib.slamhip_last_error())
    assert lib.slamhip_pair_loss_workspace_bytes(64, 1) == 64 * 128
    assert lib.slamhip_pair_loss_workspace_bytes(3, 4) == 3 * 4 * 128


def test_host_pose_entry_points_against_reference_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    T_a = fp.quat_to_transformation_batch(g["gt_a"])
    T_b = fp.quat_to_transformation_batch(g["gt_b"])
    assert np.array_equal(T_a, g["T_a"]) and np.array_equal(T_b, g["T_b"])
    np.testing.assert_allclose(fp.inverse_trans_batch(g["T_a"]), g["inv_a"], rtol=1e-15, atol=1e-16)
    rel, tw = fp.pose_batch(g["gt_a"], g["gt_b"])
    np.testing.assert_allclose(rel, g["rel"], rtol=1e-14, atol=1e-15)
    np.testing.assert_allclose(tw, g["twist"], rtol=1e-11, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(fp.trans_to_twist_batch(g["rel"], clamp=True), g["twist_clamp"],
                               rtol=1e-11, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(fp.trans_to_twist_batch(g["T_b"], clamp=True), g["abs_twist_b"],
                               rtol=1e-11, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(fp.twist_to_trans_batch(g["twist"]), g["twist_to_trans"],
                               rtol=1e-12, atol=1e-14, equal_nan=True)
    np.testing.assert_allclose(fp.twist_to_trans_batch(g["twist"], guard=False), g["twist_to_trans_noguard"],
                               rtol=1e-12, atol=1e-14, equal_nan=True)
    # the reference's guard cases: identity rotation -> NaN, out-of-range trace -> zeros
    assert np.isnan(tw[7]).all() and not tw[5].any() and not tw[6].any()


def test_reference_named_host_functions(golden_dir):
    from slam_amd.preprocess import geometry as geo
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    for i in range(g["gt_a"].shape[0]):
        T = geo._quat_to_transformation(g["gt_a"][i])
        assert np.array_equal(T, g["T_a"][i])
        np.testing.assert_allclose(geo._inverse_trans(T), g["inv_a"][i], rtol=1e-15, atol=1e-16)
        np.testing.assert_allclose(np.asarray(geo._trans_to_twist(g["rel"][i])).ravel(), g["twist"][i], rtol=1e-11, atol=1e-13,
                                   equal_nan=True)
        np.testing.assert_allclose(geo._twist_to_trans(g["twist"][i]), g["twist_to_trans"][i], rtol=1e-12,
                                   atol=1e-14, equal_nan=True)
    stamps = np.array(["1.0", "2.0", "3.5"])
    assert geo._find_label(stamps, "2.0") == 1 and geo._find_label(stamps, "2.5") == 2


def test_missing_library_is_a_hard_error(monkeypatch):
    monkeypatch.setattr(capi, "_lib", None)
    monkeypatch.setattr(capi, "LIB_PATH", "/nonexistent/libslamhip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        capi.load()
