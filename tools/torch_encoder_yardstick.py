"""The same encoder training step (12 convolutions with batch-norm and ReLU, 5 max-pools, three dense layers, weighted L2
loss, Adam; batch 128, 224x224x4) written with torch.nn on the GPU - MIOpen, hipBLASLt and torch's own element-wise kernels,
bf16 autocast, channels-last - as a yardstick for the hand-written step.  Not part of the product, not used by any test."""
import torch
import torch.nn as nn

cfg = [64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]
layers, cin = [], 4
for v in cfg:
    if v == "M":
        layers.append(nn.MaxPool2d(2))
    else:
        layers += [nn.Conv2d(cin, v, 3, padding=1, bias=False), nn.BatchNorm2d(v, eps=1e-3), nn.ReLU(inplace=True)]
        cin = v
model = nn.Sequential(*layers, nn.Flatten(), nn.Linear(25088, 4096), nn.ReLU(True), nn.Linear(4096, 4096), nn.ReLU(True),
                      nn.Linear(4096, 6)).cuda().to(memory_format=torch.channels_last)
opt = torch.optim.Adam(model.parameters(), lr=1e-5, fused=True)
x = torch.randn(128, 4, 224, 224, device="cuda").contiguous(memory_format=torch.channels_last)
gt = torch.randn(128, 6, device="cuda") * 0.1


import threading, time, sys


def heartbeat():      # the first step is silent for minutes while MIOpen compiles
    while True:
        time.sleep(30)
        print("...", flush=True)


threading.Thread(target=heartbeat, daemon=True).start()


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
    loss = ((out.float() - gt) ** 2).sum() * 0.5
    loss.backward()
    opt.step()
    return loss


for k in range(5):
    step()
    torch.cuda.synchronize()
    print("warm-up step %d done" % k, flush=True)      # (the first steps build MIOpen's kernels: minutes)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    step()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
print("torch.nn encoder training step, batch 128, bf16 autocast, channels-last: %.1f ms per step = %.0f images/s" % (ms, 128 / ms * 1e3))
