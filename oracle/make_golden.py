#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the reference's own numpy functions.

DEV-CONTAINER ONLY.  This script reads /root/reference as *text* at run time,
executes the pure-numpy function definitions it finds at fixed line ranges in a
private namespace, feeds them seeded synthetic inputs and stores
inputs + outputs as small .npz fixtures.  No reference source is written into
this repository and nothing here is imported by the shipped package, the GPU
tests, smoke() or bench.py (the reference tree does not exist on the GPU box).

Line ranges used (1-based, inclusive):
  slam/preprocess/test_import.py  14-15   _find_label
                                  18-110  pose maths
                                  148-246 geometry + _get_data
                                  255-263 intrinsics / FACTOR
  slam/network/model_input.py     14-106  pose maths (clamp variant)
                                  113-119 224x224 _overlap
                                  517-522 get_mean
  slam/preprocess/import_data.py  12-42   _absolute_position
The two Python-2 `print` statements inside those ranges are blanked before exec.

Run:  python oracle/make_golden.py     (re-creates every fixture deterministically)
"""
import hashlib
import os
import sys
import tempfile
import warnings

import numpy as np

REF = os.environ.get("SLAM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


# sha256 of the text each extraction yields (after blanking): the fixed line ranges are only meaningful for the
# reference revision they were chosen on, so anything else is refused instead of executed
EXPECTED_SHA256 = {
    "slam/preprocess/test_import.py": "251722b335ba6850f7a06c04749bc06b2ece43bd2d71c7ccfc314b1d3053b96d",
    "slam/network/model_input.py": "2a674d00ad516444cbdf16ae4f587482f192eb6c1d07abb13ad7dd4a7027bea0",
    "slam/preprocess/import_data.py": "5797882fd569341459d799c2969f4ee0940428792c5b8a7fd21282b2716c6f92",
}


class _PathOnly(object):
    """What the extracted functions use of `os`: os.path.join, nothing else."""

    class path(object):
        join = staticmethod(os.path.join)


def _load_ranges(relpath, ranges, blank=()):
    code = _extract_ranges(relpath, ranges, blank)
    digest = hashlib.sha256(code.encode()).hexdigest()
    if digest != EXPECTED_SHA256[relpath]:
        raise SystemExit("%s: the extracted line ranges hash to %s, not to the pinned %s: the reference differs from "
                         "the revision these ranges were chosen on; nothing was executed" % (relpath, digest, EXPECTED_SHA256[relpath]))
    return code


def _extract_ranges(relpath, ranges, blank=()):
    with open(os.path.join(REF, relpath)) as f:
        lines = f.read().split("\n")
    for ln in blank:
        txt = lines[ln - 1]
        indent = txt[: len(txt) - len(txt.lstrip())]
        lines[ln - 1] = indent + "pass"
    chunks = []
    for lo, hi in ranges:
        chunks.append("\n".join(lines[lo - 1:hi]))
    return "\n\n".join(chunks)


class _FakeMisc(object):
    """Stands in for scipy.misc inside the exec'd namespace: imread() loads the
    uint16 depth frame stored as .npy next to the fake PNG name."""

    @staticmethod
    def imread(path):
        return np.load(path + ".npy")


def load_test_import():
    import bisect
    ns = {"np": np, "bisect": bisect, "os": _PathOnly, "misc": _FakeMisc}
    code = _load_ranges("slam/preprocess/test_import.py",
                        [(14, 15), (18, 110), (148, 246), (255, 263)], blank=(215,))
    exec(compile(code, "<reference:test_import>", "exec"), ns)
    return ns


def load_model_input():
    ns = {"np": np}
    code = _load_ranges("slam/network/model_input.py",
                        [(14, 106), (113, 119), (517, 522)], blank=(53,))
    exec(compile(code, "<reference:model_input>", "exec"), ns)
    return ns


def rand_unit_quat(rs):
    q = rs.normal(size=4)
    return q / np.linalg.norm(q)


def quat_mul(a, b):  # (x,y,z,w) Hamilton product
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def small_motion(rs, gt_a, max_t, max_deg):
    axis = rs.normal(size=3)
    axis /= np.linalg.norm(axis)
    ang = np.deg2rad(rs.uniform(0.2, max_deg))
    dq = np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])
    q = quat_mul(gt_a[3:7], dq)
    t = gt_a[0:3] + rs.uniform(-max_t, max_t, size=3)
    return np.concatenate([t, q])


def make_pose_cases():
    rs = np.random.RandomState(11)
    cases = []
    for _ in range(4):  # TUM-like small motions
        a = np.concatenate([rs.uniform(-2, 2, 3), rand_unit_quat(rs)])
        cases.append((a, small_motion(rs, a, 0.05, 3.0)))
    a = np.concatenate([rs.uniform(-2, 2, 3), rand_unit_quat(rs)])
    cases.append((a, np.concatenate([rs.uniform(-2, 2, 3), rand_unit_quat(rs)])))  # large rotation
    # trace guard: non-unit quaternion (|q| > 1), same orientation -> (tr-1)/2 > 1
    cases.append((np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.1, 0.2, 0.3, 0.0, 0.0, 0.0, 1.01])))
    # rotation by pi about x: symmetric R -> ||w|| == 0 guard
    cases.append((np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([1.0, 2.0, 3.0, 1.0, 0.0, 0.0, 0.0])))
    # exact identity rotation -> NaN twist
    cases.append((np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.5, -0.25, 0.125, 0.0, 0.0, 0.0, 1.0])))
    # |q|>1 and rotation by pi: argument < -1 (NaN in test_import, clamp in model_input)
    cases.append((np.array([0, 0, 0, 0, 0, 0, 1.0]), np.array([0.3, 0.1, -0.2, 1.002, 0.0, 0.0, 0.0])))
    return cases


def gen_pose(ti, mi):
    cases = make_pose_cases()
    n = len(cases)
    gt_a = np.zeros((n, 7), np.float32)
    gt_b = np.zeros((n, 7), np.float32)
    T_a = np.zeros((n, 4, 4)); T_b = np.zeros((n, 4, 4)); inv_a = np.zeros((n, 4, 4))
    rel = np.zeros((n, 4, 4)); twist = np.zeros((n, 6)); twist_clamp = np.zeros((n, 6))
    twist_shape = np.zeros((n, 2), np.int64)
    exp_rel = np.zeros((n, 4, 4)); exp_mi = np.zeros((n, 4, 4))
    abs_twist_b = np.zeros((n, 6))
    for i, (a, b) in enumerate(cases):
        gt_a[i] = a.astype(np.float32)
        gt_b[i] = b.astype(np.float32)
        # the reference casts the ground-truth row to float32 before the maths
        T_a[i] = ti["_quat_to_transformation"](gt_a[i])
        T_b[i] = ti["_quat_to_transformation"](gt_b[i])
        assert np.array_equal(T_a[i], mi["_quat_to_transformation"](gt_a[i]))
        inv_a[i] = ti["_inverse_trans"](T_a[i])
        assert np.array_equal(inv_a[i], mi["_inverse_trans"](T_a[i]))
        rel[i] = np.dot(inv_a[i], T_b[i])
        tw = ti["_trans_to_twist"](rel[i])
        twist_shape[i] = (tw.ndim, tw.size)
        twist[i] = np.asarray(tw).reshape(6)
        twist_clamp[i] = np.asarray(mi["_trans_to_twist"](rel[i])).reshape(6)
        abs_twist_b[i] = np.asarray(mi["_trans_to_twist"](T_b[i])).reshape(6)
        exp_rel[i] = ti["_twist_to_trans"](twist[i])
        exp_mi[i] = mi["_twist_to_trans"](twist[i])
    np.savez_compressed(os.path.join(OUT, "pose.npz"), gt_a=gt_a, gt_b=gt_b, T_a=T_a, T_b=T_b,
                        inv_a=inv_a, rel=rel, twist=twist, twist_clamp=twist_clamp,
                        twist_ndim_size=twist_shape, twist_to_trans=exp_rel,
                        twist_to_trans_noguard=exp_mi, abs_twist_b=abs_twist_b)
    return rel


def synth_depth(rs, h, w, zero_frac=0.15):
    d = rs.randint(2500, 25001, size=(h, w)).astype(np.uint16)
    d[rs.uniform(size=(h, w)) < zero_frac] = 0
    return d


def gen_geometry(ti, rel_cases):
    rs = np.random.RandomState(5)
    out = {}
    idx = 0
    for (h, w) in ((6, 8), (48, 64)):
        for ds in (1, 2, 3):
            depth = synth_depth(rs, h, w)
            for ri in ((0, 4) if h == 6 else ((0,) if ds != 2 else (4,))):
                T = rel_cases[ri]
                pc = ti["_point_cloud"](depth, ds)
                tp = ti["_transform_pointcloud"](pc, T)
                uvd = ti["_backtransform_pointcloud"](tp, ds)
                # full-size window (640x480) keeps every point of the tiny frame in view ...
                ov_full = ti["_overlap"](uvd, [640, 480])
                # ... the frame's own size exercises both sides of the inclusive predicate
                dim = [w * 0.5, h * 1.5]
                ov_dim = ti["_overlap"](uvd, dim)
                mask = np.array([(0 <= uvd[i, 0] <= dim[0]) and (0 <= uvd[i, 1] <= dim[1])
                                 for i in range(uvd.shape[0])], dtype=np.uint8)
                assert abs(mask.sum() / float(uvd.shape[0]) - ov_dim) == 0.0
                k = "g%d_" % idx
                out[k + "depth"] = depth
                out[k + "dataset"] = np.int64(ds)
                out[k + "rel"] = T
                out[k + "pc"] = pc
                out[k + "tp"] = tp
                out[k + "uvd"] = uvd
                out[k + "dim"] = np.array(dim)
                out[k + "mask_dim"] = mask
                out[k + "overlap_dim"] = np.float64(ov_dim)
                out[k + "overlap_640x480"] = np.float64(ov_full)
                idx += 1
    out["n_cases"] = np.int64(idx)
    np.savez_compressed(os.path.join(OUT, "geometry_small.npz"), **out)


def gen_geometry_full(ti, rel_cases):
    """One 480x640 frame per intrinsics set: counts, overlap, SHA-256 of the mask
    bytes and of the float64 coordinate bytes, plus 1000 sampled rows."""
    out = {}
    for ds in (1, 2, 3):
        rs = np.random.RandomState(0 if ds == 1 else 100 + ds)
        depth = synth_depth(rs, 480, 640)
        T = rel_cases[ds - 1]
        pc = ti["_point_cloud"](depth, ds)
        tp = ti["_transform_pointcloud"](pc, T)
        uvd = ti["_backtransform_pointcloud"](tp, ds)
        ov = ti["_overlap"](uvd, [640, 480])
        mask = ((0 <= uvd[:, 0]) & (uvd[:, 0] <= 640) & (0 <= uvd[:, 1]) & (uvd[:, 1] <= 480)).astype(np.uint8)
        assert mask.sum() / float(uvd.shape[0]) == ov
        sel = np.sort(np.random.RandomState(9).choice(pc.shape[0], 1000, replace=False))
        k = "f%d_" % ds
        out[k + "seed"] = np.int64(0 if ds == 1 else 100 + ds)
        out[k + "rel"] = T
        out[k + "n_valid"] = np.int64(pc.shape[0])
        out[k + "n_inb"] = np.int64(mask.sum())
        out[k + "overlap"] = np.float64(ov)
        out[k + "mask_sha256"] = np.frombuffer(hashlib.sha256(mask.tobytes()).digest(), np.uint8)
        out[k + "pc_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(pc).tobytes()).digest(), np.uint8)
        out[k + "tp_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(tp).tobytes()).digest(), np.uint8)
        out[k + "uvd_sha256"] = np.frombuffer(hashlib.sha256(np.ascontiguousarray(uvd).tobytes()).digest(), np.uint8)
        out[k + "depth_sha256"] = np.frombuffer(hashlib.sha256(depth.tobytes()).digest(), np.uint8)
        out[k + "sel"] = sel
        out[k + "pc_sel"] = pc[sel]
        out[k + "tp_sel"] = tp[sel]
        out[k + "uvd_sel"] = uvd[sel]
    np.savez_compressed(os.path.join(OUT, "geometry_full.npz"), **out)


def gen_mean(mi):
    rs = np.random.RandomState(3)
    imgs = [rs.uniform(0, 255, size=(4, 5, 4)).astype(np.float32) for _ in range(3)]
    mean = mi["get_mean"](imgs, channels=4)
    np.savez_compressed(os.path.join(OUT, "mean.npz"), imgs=np.stack(imgs), mean=mean)
    pts = np.array([[0.0, 0.0, 1], [224.0, 224.0, 1], [224.0000001, 3, 1], [-1e-9, 3, 1], [5, np.nan, 1], [7, 9, 1]])
    np.savez_compressed(os.path.join(OUT, "overlap224.npz"), pts=pts, overlap=np.float64(mi["_overlap"](pts)))


def gen_get_data(ti):
    """Drive the reference's _get_data on a synthetic TUM-shaped sequence, with and
    without the overlap-based frame drop."""
    rs = np.random.RandomState(21)
    n_frames, h, w = 12, 24, 32
    tmp = tempfile.mkdtemp(prefix="slam_golden_")
    stamps = 1305031102.175304 + np.arange(n_frames) * 0.0334
    gt_stamps = 1305031102.1758 + np.arange(n_frames * 4) * 0.01
    gt = np.zeros((n_frames * 4, 7))
    cur = np.concatenate([rs.uniform(-1, 1, 3), rand_unit_quat(rs)])
    for i in range(gt.shape[0]):
        gt[i] = cur
        cur = small_motion(rs, cur, 0.01, 1.5)
    gt_str = np.array([["%.4f" % gt_stamps[i]] + ["%.4f" % x for x in gt[i]] for i in range(gt.shape[0])])
    assoc = np.array([["%.6f" % stamps[i], "rgb/%.6f.png" % stamps[i],
                       "%.6f" % (stamps[i] + 0.003), "depth/%.6f.png" % (stamps[i] + 0.003)]
                      for i in range(n_frames)])
    os.makedirs(os.path.join(tmp, "depth"))
    depths = np.zeros((n_frames, h, w), np.uint16)
    for i in range(n_frames):
        depths[i] = synth_depth(rs, h, w)
        np.save(os.path.join(tmp, assoc[i, 3]) + ".npy", depths[i])
    ti["filename"] = tmp
    out = dict(assoc=assoc, gt=gt_str, depths=depths)
    # (a) plain relative twists, no drop, no crop
    tw, rgbp, depp = ti["_get_data"](assoc, gt_str, 1000)
    out["a_twist"], out["a_rgb"], out["a_depth"] = tw, rgbp, depp
    # (b) static drop 2 + crop to 4 (np.random start point)
    np.random.seed(77)
    tw, rgbp, depp = ti["_get_data"](assoc, gt_str, 4, 0.0, 2)
    out["b_twist"], out["b_rgb"], out["b_depth"] = tw, rgbp, depp
    out["b_seed"] = np.int64(77)
    # (c) dynamic drop: the reference hard-codes fr1 intrinsics and a 640x480 window
    for tag, thr in (("c", 1.0), ("d", 0.9999)):
        tw, rgbp, depp = ti["_get_data"](assoc, gt_str, 1000, thr, 1)
        out[tag + "_twist"], out[tag + "_rgb"], out[tag + "_depth"] = tw, rgbp, depp
        out[tag + "_thr"] = np.float64(thr)
    np.savez_compressed(os.path.join(OUT, "get_data.npz"), **out)


def gen_abs_position():
    """slam/preprocess/import_data.py:12-42 (_absolute_position) on seeded poses."""
    ns = {"np": np}
    exec(compile(_load_ranges("slam/preprocess/import_data.py", [(12, 42)]), "<reference:import_data>", "exec"), ns)
    rs = np.random.RandomState(31)
    gt = np.stack([np.concatenate([rs.uniform(-2, 2, 3), rand_unit_quat(rs)]) for _ in range(16)])
    gt[-1, 3:] = [0.0, 0.0, 0.0, 1.0]        # identity orientation as the reference reads it: NaN / inf angles
    pos = np.stack([ns["_absolute_position"](row) for row in gt])
    np.savez_compressed(os.path.join(OUT, "abs_position.npz"), gt=gt, pos=pos)


def main():
    warnings.simplefilter("ignore")
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the dev container")
    os.makedirs(OUT, exist_ok=True)
    ti = load_test_import()
    mi = load_model_input()
    rel = gen_pose(ti, mi)
    gen_geometry(ti, rel)
    gen_geometry_full(ti, rel)
    gen_mean(mi)
    gen_get_data(ti)
    gen_abs_position()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
