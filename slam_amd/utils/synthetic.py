"""Seeded synthetic TUM-shaped RGB-D frame pairs (numpy, host side).

depth  uint16 [B,H,W]   uniform 2500..25000 raw units (0.5-5 m at 5000/m) with a
                        fraction of zeros ("noise"), or a tilted plane with mild
                        texture ("plane"), which is what real scenes look like
rgb    uint8  [B,H,W,3] uniform noise or a smooth pattern
gt     float32 [B,2,7]  pose a = random, pose b = a composed with a small motion
                        (|dt| <= max_t metres, rotation <= max_deg degrees), the
                        scale of consecutive freiburg1/xyz frames
"""
import numpy as np


def _quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def pose_pairs(n, seed=0, max_t=0.05, max_deg=3.0):
    rs = np.random.RandomState(seed)
    gt = np.zeros((n, 2, 7), np.float32)
    for i in range(n):
        q = rs.normal(size=4)
        q /= np.linalg.norm(q)
        t = rs.uniform(-2, 2, 3)
        axis = rs.normal(size=3)
        axis /= np.linalg.norm(axis)
        ang = np.deg2rad(rs.uniform(0.1, max_deg))
        dq = np.concatenate([axis * np.sin(ang / 2), [np.cos(ang / 2)]])
        gt[i, 0] = np.concatenate([t, q])
        gt[i, 1] = np.concatenate([t + rs.uniform(-max_t, max_t, 3), _quat_mul(q, dq)])
    return gt


def depth_frames(n, h, w, seed=0, kind="noise", zero_frac=0.15):
    rs = np.random.RandomState(seed)
    if kind == "noise":
        d = rs.randint(2500, 25001, size=(n, h, w)).astype(np.uint16)
    else:
        yy, xx = np.mgrid[0:h, 0:w]
        d = np.zeros((n, h, w), np.uint16)
        for i in range(n):
            a, b = rs.uniform(-4, 4, 2)
            base = rs.uniform(6000, 15000)
            plane = base + a * (xx - w / 2) + b * (yy - h / 2) + rs.normal(0, 8, size=(h, w))
            d[i] = np.clip(plane, 2500, 25000).astype(np.uint16)
    if zero_frac > 0:
        d[rs.uniform(size=d.shape) < zero_frac] = 0
    return d


def rgb_frames(n, h, w, seed=0, kind="noise"):
    rs = np.random.RandomState(seed + 7919)
    if kind == "noise":
        return rs.randint(0, 256, size=(n, h, w, 3)).astype(np.uint8)
    yy, xx = np.mgrid[0:h, 0:w]
    out = np.zeros((n, h, w, 3), np.uint8)
    for i in range(n):
        ph = rs.uniform(0, 6.28, 3)
        fr = rs.uniform(0.01, 0.05, 3)
        for c in range(3):
            out[i, ..., c] = (127.5 + 127.5 * np.sin(fr[c] * xx + 0.7 * fr[c] * yy + ph[c])).astype(np.uint8)
    return out
