#!/usr/bin/env python3
"""Kernel-only timing of slamhip_overlap_fused (64 pairs of 640x480 per call by default; B, REPS, DEPTH override)
beside the materialising path it replaces (slamhip_backproject + slamhip_reproject with counts only)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import frame_pairs as fp
from slam_amd.utils import synthetic as syn


def timed(fn, reps):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(reps):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) * 1e3 / reps)
    return min(best)


def main():
    B, H, W = int(os.environ.get("B", 64)), 480, 640
    reps = int(os.environ.get("REPS", 200))
    K = fp.intrinsics(1)
    gt = torch.from_numpy(syn.pose_pairs(B, seed=3)).cuda()
    rel = fp.relative_pose(gt[:, 0].contiguous(), gt[:, 1].contiguous())
    nbuf = 16      # 16 x 39 MB of depth: larger than the 256 MB Infinity Cache
    host = syn.depth_frames(B, H, W, seed=1, kind=os.environ.get("DEPTH", "plane"))
    bufs = [torch.from_numpy(np.roll(host, i, axis=0)).cuda() for i in range(nbuf)]
    out = {}
    us = timed(lambda i=0: out.__setitem__("c", fp.overlap_fused(bufs[i % nbuf], rel, K, (640.0, 480.0), 1.0)), reps)
    c = fp.overlap_fused(bufs[0], rel, K, (640.0, 480.0), 1.0).cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    print("overlap_fused: %.1f us per %d pairs = %.0f k pairs/s, %.0f GB/s algorithmic (2 B per pixel); counts[0]=%s"
          % (us, B, B / us * 1e3, B * H * W * 2 / us / 1e3, c[0].tolist()))

    def chain(i=0):
        xyz, valid0 = fp.backproject(bufs[i % nbuf], K, 1.0, torch.float64)
        out["m"] = fp.reproject(xyz, rel, K, (640, 480), valid0=valid0, want=("counts",))["counts"]
    us2 = timed(chain, max(5, reps // 10))
    chain(0)
    m = out["m"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    print("backproject + reproject(counts): %.1f us per %d pairs = %.0f k pairs/s; identical counts: %s"
          % (us2, B, B / us2 * 1e3, bool(np.array_equal(c, m))))


if __name__ == "__main__":
    main()
