#!/usr/bin/env python3
"""Phase timeline of the fused kernel's workgroups from in-kernel clock stamps (needs the diagnostic
build: tools/build_variant.sh stamps -DSLAMHIP_STAMPS; run with SLAMHIP_LIB=build/variants/stamps.so)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from slam_amd import capi, frame_pairs as fp
from slam_amd.utils import synthetic as syn

B, H, W = 64, 480, 640
K = fp.intrinsics(1)
gt = torch.from_numpy(syn.pose_pairs(B, seed=3)).cuda()
rel = fp.relative_pose(gt[:, 0].contiguous(), gt[:, 1].contiguous()).reshape(B, 1, 4, 4)
x = [torch.from_numpy(a).cuda() for a in (syn.depth_frames(B, H, W, seed=1, kind="plane"), syn.rgb_frames(B, H, W, seed=1),
                                           syn.depth_frames(B, H, W, seed=2, kind="plane"), syn.rgb_frames(B, H, W, seed=2))]
lib = capi.load()
nwg = 16384   # at least the launch's workgroups (9 600 on 128 x 16 tiles)
stamps = torch.zeros((nwg, 16), dtype=torch.int64, device="cuda")
lib.slamhip_debug_set_stamps.argtypes = [ctypes.c_void_p]
lib.slamhip_debug_set_stamps(None)
for _ in range(50):
    fp.pair_loss(x[0], x[1], x[2], x[3], rel, K, 1.0)
torch.cuda.synchronize()
lib.slamhip_debug_set_stamps(stamps.data_ptr())
fp.pair_loss(x[0], x[1], x[2], x[3], rel, K, 1.0)
torch.cuda.synchronize()
s = stamps.cpu().numpy()
s = s[s[:, 0] != 0]
names = ["start", "anchors issued", "window predicted", "barrier 1", "staging issued", "staging landed", "cells written",
         "barrier 2", "phase 1 done", "phase 2 done", "reduce barrier", "ticket"]
ghz = np.median((s[:, 11] - s[:, 0]) / np.maximum(s[:, 14] - s[:, 13], 1)) * 0.1
print("shader clock from s_memtime / s_memrealtime (100 MHz): %.3f GHz; kernel span by the real-time clock %.1f us"
      % (ghz, (s[:, 14].max() - s[:, 13].min()) / 100.0))
t0 = s[:, 0].min()
print("workgroups stamped: %d; kernel span %.1f us (ticks are shader cycles; 2.1 GHz assumed)" % (len(s), (s[:, 11].max() - t0) / 2100.0))
d = (s[:, 12] - s[:, 1]) / 2100.0
print("%-18s median %.2f us  mean %.2f  p90 %.2f (after the anchor loads were issued)" % ("pose arrived", np.median(d), d.mean(), np.percentile(d, 90)))
if (s[:, 15] != 0).any():
    d = (s[:, 15] - s[:, 0]) / 2100.0
    print("%-18s median %.2f us  mean %.2f  p90 %.2f (pose waited for right after its load, before the anchors)" % ("pose alone", np.median(d), d.mean(), np.percentile(d, 90)))
prev = s[:, 0]
for i in range(1, 12):
    d = (s[:, i] - prev) / 2100.0
    print("%-18s median %.2f us  mean %.2f  p90 %.2f" % (names[i], np.median(d), d.mean(), np.percentile(d, 90)))
    prev = s[:, i]
life = (s[:, 11] - s[:, 0]) / 2100.0
print("lifetime           median %.2f us  mean %.2f  p90 %.2f" % (np.median(life), life.mean(), np.percentile(life, 90)))
# residency over the launch (real-time clock, 10 ns ticks): workgroups resident per 4 us bin, and the mean
# lifetime of the workgroups that started in that bin
r0 = s[:, 13].min()
st, en = (s[:, 13] - r0) / 100.0, (s[:, 14] - r0) / 100.0
span = en.max()
print("bin start us | resident workgroups (mean) | started | mean lifetime of those us")
for a in np.arange(0.0, span, 4.0):
    b = a + 4.0
    resident = np.clip(np.minimum(en, b) - np.maximum(st, a), 0, None).sum() / 4.0
    started = (st >= a) & (st < b)
    print("%8.0f | %8.1f | %5d | %s" % (a, resident, started.sum(), "%.2f" % (en[started] - st[started]).mean() if started.any() else "-"))
# the fill of the launch in 0.5 us bins: workgroups started and finished so far
print("fill: time us | started so far | finished so far")
for a in np.arange(0.5, 12.01, 0.5):
    print("%6.1f | %5d | %5d" % (a, (st < a).sum(), (en < a).sum()))
