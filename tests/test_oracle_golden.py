"""The numpy oracle against golden vectors produced by the reference's own
functions (tests/golden/*.npz, made by oracle/make_golden.py)."""
import hashlib
import os

import numpy as np
import pytest

from oracle import slam_oracle as so


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def synth_depth(rs, h, w, zero_frac=0.15):
    d = rs.randint(2500, 25001, size=(h, w)).astype(np.uint16)
    d[rs.uniform(size=(h, w)) < zero_frac] = 0
    return d


def test_fma_is_exact():
    from fractions import Fraction
    rs = np.random.RandomState(0)
    a = rs.uniform(-1e4, 1e4, 4000)
    b = rs.uniform(-1, 1, 4000)
    c = -a * b * (1 + rs.uniform(-1e-15, 1e-15, 4000))  # heavy cancellation
    c[::2] = rs.uniform(-1e4, 1e4, 2000)
    got = so.fma(a, b, c)
    for i in range(a.size):
        want = float(Fraction(a[i]) * Fraction(b[i]) + Fraction(c[i]))
        assert got[i] == want


def test_transform_matches_blas_dot_here():
    # informative: on x86 BLAS builds with FMA micro-kernels np.dot equals the fma chain
    rs = np.random.RandomState(1)
    P = rs.uniform(-1e4, 1e4, (5000, 3))
    T = np.eye(4)
    T[:3, :3] = rs.uniform(-1, 1, (3, 3))
    T[:3, 3] = rs.uniform(-1, 1, 3)
    ref = np.dot(P, T[:3, :3]) + T[:3, 3]
    got = so.transform_pointcloud(P, T)
    np.testing.assert_allclose(got, ref, rtol=4e-16, atol=0)


def test_pose_golden(golden_dir):
    g = _load(golden_dir, "pose.npz")
    n = g["gt_a"].shape[0]
    for i in range(n):
        Ta = so.quat_to_transformation(g["gt_a"][i])
        Tb = so.quat_to_transformation(g["gt_b"][i])
        assert np.array_equal(Ta, g["T_a"][i])
        assert np.array_equal(Tb, g["T_b"][i])
        inv = so.inverse_trans(Ta)
        np.testing.assert_allclose(inv, g["inv_a"][i], rtol=1e-15, atol=1e-16)
        rel = so.relative_pose(Ta, Tb)
        np.testing.assert_allclose(rel, g["rel"][i], rtol=1e-14, atol=1e-15)
        tw = so.trans_to_twist(g["rel"][i])
        np.testing.assert_allclose(tw, g["twist"][i], rtol=1e-11, atol=1e-13, equal_nan=True)
        twc = so.trans_to_twist(g["rel"][i], clamp=True)
        np.testing.assert_allclose(twc, g["twist_clamp"][i], rtol=1e-11, atol=1e-13, equal_nan=True)
        tab = so.trans_to_twist(g["T_b"][i], clamp=True)
        np.testing.assert_allclose(tab, g["abs_twist_b"][i], rtol=1e-11, atol=1e-13, equal_nan=True)
        np.testing.assert_allclose(so.twist_to_trans(g["twist"][i]), g["twist_to_trans"][i],
                                   rtol=1e-12, atol=1e-14, equal_nan=True)
        np.testing.assert_allclose(so.twist_to_trans(g["twist"][i], guard=False), g["twist_to_trans_noguard"][i],
                                   rtol=1e-12, atol=1e-14, equal_nan=True)
    # guard cases really are in the fixture
    assert np.isnan(g["twist"][7]).all() and not g["twist"][5].any() and not g["twist"][6].any()
    assert np.isnan(g["twist"][8]).all() and not g["twist_clamp"][8].any()


def test_geometry_small_golden(golden_dir):
    g = _load(golden_dir, "geometry_small.npz")
    for i in range(int(g["n_cases"])):
        k = "g%d_" % i
        ds = int(g[k + "dataset"])
        depth, rel = g[k + "depth"], g[k + "rel"]
        pc = so.point_cloud(depth, ds)
        assert np.array_equal(pc, g[k + "pc"])                      # bit-exact
        tp = so.transform_pointcloud(pc, rel)
        assert np.array_equal(tp, g[k + "tp"])                      # bit-exact (fma chain == dgemm)
        uvd = so.backtransform_pointcloud(tp, so.intrinsics(ds))
        assert np.array_equal(uvd, g[k + "uvd"])
        dim = g[k + "dim"]
        assert np.array_equal(so.inbounds_mask(uvd, dim).astype(np.uint8), g[k + "mask_dim"])
        assert so.overlap(uvd, dim) == float(g[k + "overlap_dim"])
        assert so.overlap(uvd, [640, 480]) == float(g[k + "overlap_640x480"])
        # the dense form compacts to the reference's arrays
        P = so.project_dense(depth, rel, so.intrinsics(ds))
        v0 = P["valid0"].reshape(-1)
        assert np.array_equal(P["xyz"].reshape(-1, 3)[v0], g[k + "pc"])
        assert np.array_equal(P["uvd"].reshape(-1, 3)[v0], g[k + "uvd"])


def test_geometry_full_golden(golden_dir):
    g = _load(golden_dir, "geometry_full.npz")
    for ds in (1, 2, 3):
        k = "f%d_" % ds
        depth = synth_depth(np.random.RandomState(int(g[k + "seed"])), 480, 640)
        assert np.array_equal(_sha(depth), g[k + "depth_sha256"])
        P = so.project_dense(depth, g[k + "rel"], so.intrinsics(ds))
        v0 = P["valid0"].reshape(-1)
        pc = P["xyz"].reshape(-1, 3)[v0]
        tp = P["xyz2"].reshape(-1, 3)[v0]
        uvd = P["uvd"].reshape(-1, 3)[v0]
        assert pc.shape[0] == int(g[k + "n_valid"])
        assert np.array_equal(_sha(pc), g[k + "pc_sha256"])
        assert np.array_equal(_sha(tp), g[k + "tp_sha256"])
        assert np.array_equal(_sha(uvd), g[k + "uvd_sha256"])
        mask = so.inbounds_mask(uvd, [640, 480]).astype(np.uint8)
        assert int(mask.sum()) == int(g[k + "n_inb"])
        assert np.array_equal(_sha(mask), g[k + "mask_sha256"])
        assert so.overlap(uvd, [640, 480]) == float(g[k + "overlap"])
        assert int(P["inb"].sum()) == int(g[k + "n_inb"])
        sel = g[k + "sel"]
        assert np.array_equal(uvd[sel], g[k + "uvd_sel"])


def test_mean_and_overlap224_golden(golden_dir):
    g = _load(golden_dir, "mean.npz")
    got = so.get_mean(list(g["imgs"]), channels=4)
    assert np.array_equal(got, g["mean"])
    o = _load(golden_dir, "overlap224.npz")
    assert so.overlap_224(o["pts"]) == float(o["overlap"])


def test_get_data_golden(golden_dir):
    g = _load(golden_dir, "get_data.npz")
    assoc, gt, depths = g["assoc"], g["gt"], g["depths"]
    index = {assoc[i, 3]: i for i in range(assoc.shape[0])}
    tw, rp, dp = so.get_data(assoc, gt, 1000)
    np.testing.assert_allclose(tw, g["a_twist"], rtol=1e-11, atol=1e-13)
    assert list(rp) == list(g["a_rgb"]) and list(dp) == list(g["a_depth"])
    rng = np.random.RandomState(int(g["b_seed"]))
    tw, rp, dp = so.get_data(assoc, gt, 4, 0.0, 2, rng=rng)
    np.testing.assert_allclose(tw, g["b_twist"], rtol=1e-11, atol=1e-13)
    assert list(rp) == list(g["b_rgb"]) and list(dp) == list(g["b_depth"])
    for tag in ("c", "d"):
        tw, rp, dp = so.get_data(assoc, gt, 1000, float(g[tag + "_thr"]), 1,
                                 read_depth=lambda p: depths[index[p]])
        np.testing.assert_allclose(tw, g[tag + "_twist"], rtol=1e-11, atol=1e-13)
        assert list(rp) == list(g[tag + "_rgb"]) and list(dp) == list(g[tag + "_depth"])
    # bisect beyond the last stamp is an IndexError, as in the reference
    bad = assoc.copy()
    bad[0, 0] = "9" * 10
    with pytest.raises(IndexError):
        so.get_data(bad, gt, 1000)


def test_losses():
    rs = np.random.RandomState(2)
    o, g = rs.normal(size=(5, 6)), rs.normal(size=(5, 6))
    assert so.l2_loss_weighted(o, g, np.zeros((6, 6))) == 0.0
    assert abs(so.l2_loss_weighted(o, g, np.eye(6)) - 0.5 * np.sum((o - g) ** 2)) < 1e-12
    assert abs(so.lstm_loss(o, g, np.eye(6)) - np.sum((o - g) ** 2) / 5) < 1e-12
    o7, g7 = rs.normal(size=(3, 7)), rs.normal(size=(3, 7))
    want = np.sum((o7[:, :3] - g7[:, :3]) ** 2) + 1100 * np.sum((o7[:, 3:] - g7[:, 3:]) ** 2)
    assert abs(so.posenet_loss(o7, g7) - want) < 1e-9


def test_absolute_position_golden(golden_dir):
    """slam/preprocess/import_data.py:12-42: the restatement and the library's host function
    against the reference function's own outputs."""
    g = _load(golden_dir, "abs_position.npz")
    want = g["pos"]
    got = np.stack([so.absolute_position(row) for row in g["gt"]])
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-14, equal_nan=True)
    from slam_amd import frame_pairs as fp
    from slam_amd.preprocess import import_data
    np.testing.assert_allclose(fp.absolute_position_batch(g["gt"]), want, rtol=1e-11, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(import_data._absolute_position(g["gt"][0]), want[0], rtol=1e-11, atol=1e-13)
    rgb = np.arange(48 * 64 * 3, dtype=np.uint8).reshape(48, 64, 3)
    depth = (np.arange(48 * 64).reshape(48, 64, 1) * 7 + 500).astype(np.uint16)
    img = import_data._image_preprocessing(rgb, depth)
    assert img.shape == (224, 224, 4) and abs(img[..., 3].max() - 255.0) < 1.0 and abs(img[..., 3].min()) < 1.0
