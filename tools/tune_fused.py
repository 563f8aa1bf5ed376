#!/usr/bin/env python3
"""Kernel-only timing of slamhip_pair_loss_fused for tuning (set SLAMHIP_LIB to
try another build of the library).  Prints microseconds per launch (B anchors, default 64; KN neighbours per
anchor; SIZE=HxW; DEPTH=plane|noise; FACTOR)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import capi, frame_pairs as fp
from slam_amd.utils import synthetic as syn


def main():
    B = int(os.environ.get("B", 64))
    H, W = [int(v) for v in os.environ.get("SIZE", "480x640").lower().split("x")]
    k = int(os.environ.get("KN", 1))                 # neighbour frames per anchor
    depth_kind = os.environ.get("DEPTH", "plane")
    factor = float(os.environ.get("FACTOR", 1.0))
    nbuf = int(os.environ.get("NBUF", 4))
    reps = int(os.environ.get("REPS", 100))
    K = tuple(v * W / 640.0 for v in fp.intrinsics(1))
    gt = torch.from_numpy(syn.pose_pairs(B * k, seed=3)).cuda()
    rel = fp.relative_pose(gt[:, 0].contiguous(), gt[:, 1].contiguous()).reshape(B, k, 4, 4)
    host = [syn.depth_frames(B, H, W, seed=1, kind=depth_kind), syn.rgb_frames(B, H, W, seed=1, kind="noise"),
            syn.depth_frames(B * k, H, W, seed=2, kind=depth_kind).reshape(B, k, H, W),
            syn.rgb_frames(B * k, H, W, seed=2, kind="noise").reshape(B, k, H, W, 3)]
    bufs = [[torch.from_numpy(np.roll(x, i, axis=0)).cuda() for x in host] + [rel.roll(i, 0).contiguous()]
            for i in range(nbuf)]
    sums = torch.empty((B, k, 4), dtype=torch.float64, device="cuda")
    counts = torch.empty((B, k, 4), dtype=torch.int32, device="cuda")
    lib = capi.load()
    ws = torch.zeros((lib.slamhip_pair_loss_workspace_bytes(B, k),), dtype=torch.uint8, device="cuda")
    Kc = capi.K4(K)
    stream = capi.current_stream()

    def launch(x):
        rc = lib.slamhip_pair_loss_fused(x[0].data_ptr(), x[1].data_ptr(), x[2].data_ptr(), x[3].data_ptr(),
                                         x[4].data_ptr(), Kc, factor, B, k, H, W, sums.data_ptr(),
                                         counts.data_ptr(), ws.data_ptr(), ws.numel(), stream)
        assert rc == 0, lib.slamhip_last_error()

    for i in range(10):
        launch(bufs[i % nbuf])
    torch.cuda.synchronize()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        e0.record()
        for i in range(reps):
            launch(bufs[i % nbuf])
        e1.record()
        torch.cuda.synchronize()
        best.append((e0.elapsed_time(e1) * 1e3 / reps, (time.perf_counter() - t0) * 1e6 / reps))
    us = min(b[0] for b in best)
    print("%s: %.1f us/launch (gpu events), %.1f us/launch (host wall), %.0f GB/s algorithmic, counts[0]=%s"
          % (os.environ.get("SLAMHIP_LIB", "default"), us, min(b[1] for b in best),
             B * H * W * 5 * (1 + k) / us / 1e3, (counts[0, 0].cpu().numpy().astype(np.int64) & 0xFFFFFFFF).tolist()))


if __name__ == "__main__":
    main()
