"""Plain PyTorch float32 restatement of the pose-regression encoder on the CPU
(slam/network/cnn_model.py:31-80,139-140: 12 conv3x3 SAME layers with batch-norm on
batch statistics + ReLU, five 2x2 max-pools, two ReLU dense layers, linear output,
weighted L2 loss) with autograd.  TEST INFRASTRUCTURE ONLY: the checker of the HIP
kernels in tests/ and the timed CPU baseline of bench.py; the product never imports it.
Parity with the reference itself is unpinned (TensorFlow 0.x is not available)."""
import time

import torch
import torch.nn.functional as F

CONV_LAYERS = (
    ("conv1_1", 64, False), ("conv1_2", 64, True),
    ("conv2_1", 128, False), ("conv2_2", 128, True),
    ("conv3_1", 256, False), ("conv3_2", 256, True),
    ("conv4_1", 512, False), ("conv4_2", 512, False), ("conv4_3", 512, True),
    ("conv5_1", 512, False), ("conv5_2", 512, False), ("conv5_3", 512, True),
)


def forward_loss(P, x, gt, Wl, eps, storage_rounding=False):
    """P: dict of float32 tensors named like VGG16Model.params (conv HWIO weights,
    '<layer>/bn' = [beta; scale], fc [in,out]); x [B,H,W,C] float32.  storage_rounding
    rounds weights and stored activations to bfloat16 where the GPU path stores them
    (straight-through for the gradient).  Returns (output [B,D], loss)."""

    def q(t):
        return t + (t.to(torch.bfloat16).float() - t).detach() if storage_rounding else t

    h = x.permute(0, 3, 1, 2)
    for name, _, pool in CONV_LAYERS:
        h = q(F.conv2d(h, q(P[name + "/weights"]).permute(3, 2, 0, 1), padding=1))
        mean = h.mean(dim=(0, 2, 3), keepdim=True)
        var = h.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        bn = P[name + "/bn"]
        h = q(torch.relu((h - mean) * torch.rsqrt(var + eps) * bn[1].view(1, -1, 1, 1) + bn[0].view(1, -1, 1, 1)))
        if pool:
            h = F.max_pool2d(h, 2)
    h = h.permute(0, 2, 3, 1).reshape(x.shape[0], -1)     # NHWC flatten, as tf.reshape does
    h = q(torch.relu(h @ q(P["fc6-conv/weights"]) + P["fc6-conv/bias"]))
    h = q(torch.relu(h @ q(P["fc7-conv/weights"]) + P["fc7-conv/bias"]))
    D = gt.shape[1]
    out = h @ q(P["out_layer/weights"])[:, :D] + P["out_layer/bias"][:D]
    loss = 0.5 * (((out - gt) @ Wl) ** 2).sum()
    return out, loss


def random_params(height, width, in_channels=4, fc_dim=4096, output_dim=6, seed=0):
    g = torch.Generator().manual_seed(seed)
    P, c = {}, in_channels
    for name, cout, _ in CONV_LAYERS:
        P[name + "/weights"] = 0.05 * torch.randn(3, 3, c, cout, generator=g)
        P[name + "/bn"] = torch.stack([torch.zeros(cout), torch.ones(cout)])
        c = cout
    feat = (height // 32) * (width // 32) * 512
    for name, k, n in (("fc6-conv", feat, fc_dim), ("fc7-conv", fc_dim, fc_dim), ("out_layer", fc_dim, output_dim)):
        P[name + "/weights"] = 0.05 * torch.randn(k, n, generator=g)
        P[name + "/bias"] = 0.05 * torch.randn(n, generator=g)
    return P


def timed_training_steps(batch, budget_s=10.0, height=224, width=224, seed=0):
    """forward + loss + backward (autograd) on the host, float32: images/s and the
    thread count torch used."""
    P = {k: v.requires_grad_() for k, v in random_params(height, width, seed=seed).items()}
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, height, width, 4, generator=g)
    gt = 0.1 * torch.randn(batch, 6, generator=g)
    Wl = torch.eye(6)
    steps, t0 = 0, time.perf_counter()
    while True:
        for v in P.values():
            v.grad = None
        _, loss = forward_loss(P, x, gt, Wl, 1e-3)
        loss.backward()
        steps += 1
        el = time.perf_counter() - t0
        if el >= budget_s:
            break
    return dict(value=steps * batch / el, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d training steps (forward + backward, no optimizer) at batch %d in %.1f s, "
                       "oracle/torch_encoder.py, PyTorch float32 CPU" % (steps, batch, el))


def timed_forward(batch, budget_s=5.0, height=224, width=224, seed=0):
    """The forward pass alone on the host, float32 (the pose-net forward of the CPU-runnable configuration)."""
    P = random_params(height, width, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    x = torch.randn(batch, height, width, 4, generator=g)
    gt = 0.1 * torch.randn(batch, 6, generator=g)
    Wl = torch.eye(6)
    steps, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while True:
            forward_loss(P, x, gt, Wl, 1e-3)
            steps += 1
            el = time.perf_counter() - t0
            if el >= budget_s:
                break
    return dict(value=steps * batch / el, unit="images/s", cores=torch.get_num_threads(), kind="port",
                sample="%d forward passes at batch %d in %.1f s, oracle/torch_encoder.py, PyTorch float32 CPU"
                       % (steps, batch, el))
