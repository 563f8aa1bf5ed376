"""The vendor convolution (MIOpen through torch.nn.functional.conv2d, bf16, channels-last) on the encoder's layer shapes at
batch 128: forward, and forward + backward, as a yardstick for the kernels of this repository.  Not part of the product."""
import sys
import torch
import torch.nn.functional as F


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (hw, cin, cout) in ((224, 64, 64), (112, 128, 128), (56, 256, 256), (28, 512, 512), (14, 512, 512)):
    x = torch.randn(128, cin, hw, hw, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    w = torch.randn(cout, cin, 3, 3, device="cuda").to(torch.bfloat16).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    fl = 2.0 * 128 * hw * hw * cin * cout * 9
    f_ms = timed(lambda: F.conv2d(x, w, padding=1))
    y = F.conv2d(x, w, padding=1)
    g = torch.randn_like(y)

    def fb():
        yy = F.conv2d(x, w, padding=1)
        yy.backward(g)
        x.grad = None
        w.grad = None
    fb_ms = timed(fb)
    print("%dx%d %d->%d: forward %.0f us = %.0f TFLOP/s; forward + both gradients %.0f us = %.0f TFLOP/s"
          % (hw, hw, cin, cout, f_ms * 1e3, fl / f_ms / 1e9, fb_ms * 1e3, 3 * fl / fb_ms / 1e9), flush=True)
    del x, w, y, g
