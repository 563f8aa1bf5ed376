"""Times one convolution shape through the C ABI: forward (igemm), data gradient and
filter gradient.  CONV="B,H,W,Cin,Cout" REPS=n; prints us and TFLOP/s per kernel."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.network import encoder_ops as ops   # noqa: E402

B, H, W, C, N = [int(v) for v in os.environ.get("CONV", "128,28,28,512,512").split(",")]
reps = int(os.environ.get("REPS", 20))
which = os.environ.get("WHICH", "fwd,dgrad,wgrad").split(",")
bf = dict(dtype=torch.bfloat16, device="cuda")
if os.environ.get("ZERO", "0") == "1":      # all-zero operands: the same instructions at a fraction of the switching power
    x, dy, w = torch.zeros(B, H, W, C, **bf), torch.zeros(B, H, W, N, **bf), torch.zeros(3, 3, C, N, device="cuda")
else:
    x = torch.randn(B, H, W, C, **bf)
    dy = torch.randn(B, H, W, N, **bf)
    w = 0.05 * torch.randn(3, 3, C, N, device="cuda")
w_fwd, w_dg = torch.empty(N, 9 * C, **bf), torch.empty(C, 9 * N, **bf)
ops.conv_weight_layouts(w, w_fwd, w_dg)
out, dx = torch.empty(B, H, W, N, **bf), torch.empty(B, H, W, C, **bf)
dw = torch.empty(3, 3, C, N, device="cuda")
flops = 2.0 * B * H * W * 9 * C * N
runs = {"fwd": lambda: ops.conv3x3(x, w_fwd, out_bf16=out), "dgrad": lambda: ops.conv3x3(dy, w_dg, out_bf16=dx, flip=True),
        "wgrad": lambda: ops.wgrad(x, dy, dw)}
for name in which:
    f = runs[name]
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%s %s: %.1f us, %.0f TFLOP/s" % (os.environ.get("CONV", ""), name, us, flops / us / 1e6))
