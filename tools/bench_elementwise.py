"""Times the batch-norm forward / backward kernels on one layer shape.
SHAPE="B,H,W,C" (default the first layer of the encoder at batch 128), REPS."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.network import encoder_ops as ops   # noqa: E402


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


def main():
    B, H, W, C = (int(v) for v in os.environ.get("SHAPE", "128,224,224,64").split(","))
    reps = int(os.environ.get("REPS", "20"))
    dev = torch.device("cuda")
    bf = dict(dtype=torch.bfloat16, device=dev)
    z = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(B, H, W, C, device=dev).to(torch.bfloat16)
    y, dz = torch.empty_like(z), torch.empty_like(z)
    pooled = torch.empty(B, H // 2, W // 2, C, **bf)
    dp = torch.randn(B, H // 2, W // 2, C, device=dev).to(torch.bfloat16)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    sums, red = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    ops.channel_stats(z, sums)
    nbytes = z.numel() * 2
    for name, fn, passes in (
            ("bn_relu_fwd", lambda: ops.bn_relu(z, sums, gamma, beta, y, 1e-3), 2),
            ("bn_relu_bwd (reduce + apply)", lambda: ops.bn_relu_bwd(dy, z, sums, gamma, beta, red, dz, 1e-3), 5),
            ("bn_relu_pool_fwd", lambda: ops.bn_relu_pool(z, sums, gamma, beta, y, pooled, 1e-3), 2.25),
            ("bn_relu_pool_bwd (reduce + apply)", lambda: ops.bn_relu_pool_bwd(dp, z, sums, gamma, beta, red, dz, 1e-3), 3.5)):
        us = timed(fn, reps)
        print("%s %s: %.1f us, %.2f TB/s" % (os.environ.get("SHAPE", "128,224,224,64"), name, us, passes * nbytes / us / 1e6))


if __name__ == "__main__":
    main()
