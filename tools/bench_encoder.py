"""Times the encoder training step (forward + loss + backward + Adam) at the
configuration the baseline names for it: batch 128, 224x224x4 input, bfloat16
operands.  Prints one JSON line with images/s, the model FLOPs per step and the
fraction of the dense bf16 MFMA peak.  ENC_B / ENC_HW / ENC_STEPS override; ENC_GRAPH=0 launches
the step eagerly, ENC_FWD=0 leaves out the forward-only timing.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from slam_amd.network.cnn_model import VGG16Model, model_flops   # noqa: E402

MFMA_BF16_PEAK_TFLOPS = 2500.0


def main():
    B = int(os.environ.get("ENC_B", 128))
    HW = int(os.environ.get("ENC_HW", 224))
    steps = int(os.environ.get("ENC_STEPS", 5))
    dev = torch.device("cuda:0")
    x = torch.randn(B, HW, HW, 4, device=dev)
    model = VGG16Model(B, x, 6, 1e-3)
    model.build_graph()
    gt = torch.randn(B, 6, device=dev) * 0.1
    model.add_loss(torch.eye(6, device=dev), gt)
    step = model.add_optimizer()
    model._optimizer.learning_rate = 1e-5
    if os.environ.get("ENC_GRAPH", "1") != "0":     # the step replayed from a HIP graph (ENC_GRAPH=0: eager launches)
        step = model.capture_train_step(warmup=2)
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    # forward only
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fwd_steps = steps if os.environ.get("ENC_FWD", "1") != "0" else 0     # ENC_FWD=0: whole steps only (counter runs)
    e0.record()
    for _ in range(fwd_steps):
        model.forward()
    e1.record()
    torch.cuda.synchronize()
    fwd_ms = e0.elapsed_time(e1) / max(fwd_steps, 1)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(steps):
        step()
    e1.record()
    torch.cuda.synchronize()
    wall_ms = (time.perf_counter() - t0) * 1e3 / steps
    step_ms = e0.elapsed_time(e1) / steps
    f_fwd, f_all = model_flops(B, HW, HW, 4, model.fc_dim, 6)
    print(json.dumps({
        "workload": "encoder train step, batch %d, %dx%dx4, bf16 operands / f32 accumulate" % (B, HW, HW),
        "images_per_s": B / (step_ms * 1e-3), "step_ms": step_ms, "step_wall_ms": wall_ms, "forward_ms": fwd_ms,
        "tflops_step": f_all / 1e12, "achieved_tflops": f_all / 1e12 / (step_ms * 1e-3),
        "forward_achieved_tflops": f_fwd / 1e12 / (fwd_ms * 1e-3) if fwd_steps else None,
        "mfma_peak_tflops": MFMA_BF16_PEAK_TFLOPS,
        "mfma_frac": f_all / 1e12 / (step_ms * 1e-3) / MFMA_BF16_PEAK_TFLOPS,
        "hip_graph": os.environ.get("ENC_GRAPH", "1") != "0", "loss": float(model.loss), "hbm_gb_allocated": torch.cuda.max_memory_allocated() / 1e9}))


if __name__ == "__main__":
    main()
