"""Drop-in for the frame-pair geometry functions of the reference's
slam/preprocess/test_import.py (same names, argument order and return shapes),
running on the MI355X through libslamhip.

numpy arrays in, numpy arrays out, like the reference; each call moves its
arrays to the GPU, runs the HIP kernel and copies the result back, so these are
for scripts written against the reference.  Code that keeps frames resident in
HBM should use slam_amd.frame_pairs directly.
"""
import bisect

import numpy as np
import torch

from .. import frame_pairs as fp

FX, FY, CX, CY, FACTOR = list(fp.FX), list(fp.FY), list(fp.CX), list(fp.CY), 1


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("slam_amd.preprocess.geometry needs a GPU (no CPU fallback)")
    return torch.device("cuda")


def _find_label(groundtruth, timestamp):
    """test_import.py:14-15"""
    return bisect.bisect_left(groundtruth, timestamp)


def _quat_to_transformation(groundtruth):
    """test_import.py:18-47"""
    return fp.quat_to_transformation_batch(np.asarray(groundtruth, dtype=np.float32))[0]


def _inverse_trans(trans):
    """test_import.py:80-86"""
    return fp.inverse_trans_batch(trans)[0]


def _trans_to_twist(trans, clamp=False):
    """test_import.py:50-77: np.matrix[1,6] on the main path, zeros(6) on the
    guard paths (the reference's return shapes)."""
    trans = np.asarray(trans, dtype=np.float64)
    tw = fp.trans_to_twist_batch(trans, clamp)[0]
    if (np.trace(trans[0:3, 0:3]) - 1) / 2 > 1 or not tw[3:6].any():
        return np.zeros(6)
    return np.matrix(tw)


def _twist_to_trans(twist):
    """test_import.py:89-110"""
    return fp.twist_to_trans_batch(np.asarray(twist, dtype=np.float64).reshape(6))[0]


def _point_cloud(depth_image, dataset_int):
    """test_import.py:148-156: compacted float64 [N_valid,3] in row-major pixel order."""
    d = torch.from_numpy(np.ascontiguousarray(depth_image, dtype=np.uint16)[None]).to(_device())
    xyz, valid0 = fp.backproject(d, fp.intrinsics(dataset_int), FACTOR, torch.float64)
    return xyz.reshape(-1, 3)[valid0.reshape(-1).bool()].cpu().numpy()


def _as_cloud(pointcloud):
    p = np.ascontiguousarray(pointcloud, dtype=np.float64).reshape(-1, 3)
    return torch.from_numpy(p).to(_device()).reshape(1, 1, -1, 3), p.shape[0]


def _transform_pointcloud(pointcloud, trans):
    """test_import.py:159-160"""
    x, n = _as_cloud(pointcloud)
    rel = torch.from_numpy(np.ascontiguousarray(trans, dtype=np.float64).reshape(1, 4, 4)).to(x.device)
    out = fp.reproject(x, rel, (1.0, 1.0, 0.0, 0.0), (0.0, 0.0), want=("xyz2",))
    return out["xyz2"].reshape(n, 3).cpu().numpy()


def _backtransform_pointcloud(pointcloud, dataset_int):
    """test_import.py:163-168"""
    x, n = _as_cloud(pointcloud)
    out = fp.reproject(x, None, fp.intrinsics(dataset_int), (0.0, 0.0), want=("uvd",))
    return out["uvd"].reshape(n, 3).cpu().numpy()


def _overlap(backtransform, dim):
    """test_import.py:171-176"""
    u = np.ascontiguousarray(backtransform, dtype=np.float64).reshape(-1, 3)
    cnt, _ = fp.overlap_count(torch.from_numpy(u).to(_device()), dim)
    return float(int(cnt.item()) & 0xFFFFFFFF) / u.shape[0]


def _get_data(associations, groundtruth, sequence_length, dynamic_drop=0.0, static_drop=1,
              read_depth=None, rng=np.random, clamp=False):
    """test_import.py:178-246.  Labels on the host C path, the overlap test of the
    dynamic drop on the GPU in one pass over the previous frame's depth plane, which stays
    resident (slamhip_overlap_fused: the counts of :148-176 without the point cloud).
    `read_depth(path) -> uint16 [H,W]` replaces the reference's PNG read (:204).
    clamp=True takes the twist with the clamped arccos argument, the variant the
    queued provider's copy of this function calls (model_input.py:46-54, 243-319)."""
    associations = associations[0::static_drop]
    rgb_filepaths = associations[:, 1]
    depth_filepaths = associations[:, 3]
    n = associations.shape[0]
    twist = np.zeros((n, 6))
    drop = 0.0 < dynamic_drop <= 1.0
    delete_list = []
    K = fp.intrinsics(1)
    gt7 = np.stack([groundtruth[:, 1:][_find_label(groundtruth[:, 0], associations[i, 0])].astype(np.float32)
                    for i in range(n)]) if n else np.zeros((0, 7), np.float32)
    old = 0  # index of the last kept frame
    depth_old = None
    overlap = None
    for i in range(n):
        if drop:
            d = torch.from_numpy(np.ascontiguousarray(read_depth(associations[i, 3]), dtype=np.uint16)[None]).to(_device())
        if i > 0:
            rel, tw = fp.pose_batch(gt7[old:old + 1], gt7[i:i + 1], clamp)
            if drop:
                c = fp.overlap_fused(depth_old, torch.from_numpy(rel).to(depth_old.device), K, (640.0, 480.0), FACTOR)
                c = c.cpu().numpy().astype(np.int64) & 0xFFFFFFFF
                overlap = float(c[0, 1]) / float(c[0, 0])
                if overlap < dynamic_drop:
                    twist[i] = tw[0]
                else:
                    delete_list.append(i)
            else:
                twist[i] = tw[0]
        if i == 0 or (drop and overlap < dynamic_drop):
            old = i
        if drop:
            depth_old = d
    if drop:
        twist = np.delete(twist, delete_list, 0)
        rgb_filepaths = np.delete(rgb_filepaths, delete_list, 0)
        depth_filepaths = np.delete(depth_filepaths, delete_list, 0)
    n = twist.shape[0]
    sequence_length = np.min([sequence_length, n])
    start = 0
    if n > sequence_length:
        start = rng.randint(0, n - sequence_length)
    sl = slice(start, start + sequence_length)
    return twist[sl], rgb_filepaths[sl], depth_filepaths[sl]
