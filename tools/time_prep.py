#!/usr/bin/env python3
"""Kernel time of the two input-preparation entry points (slamhip_prep_rgbd_tf, slamhip_prep_rgbd) on 128 raw
640x480 frames resident in HBM."""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from slam_amd.network import prep
from slam_amd.utils import synthetic as syn
B=128
rgb=torch.from_numpy(syn.rgb_frames(B,480,640,seed=1,kind="noise")).cuda(); d=torch.from_numpy(syn.depth_frames(B,480,640,seed=2,kind="noise")).cuda()
for f,name in ((lambda: prep.rgbd_batch_tf(rgb,d,(224,224)),"tf"),(lambda: prep.rgbd_batch(rgb,d,(224,224)),"host-form")):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    print(name, "%.1f us per 128 frames" % (e0.elapsed_time(e1)/20*1e3))
