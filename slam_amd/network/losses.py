"""The other losses of the reference's trainers on the GPU (float32):
the LSTM trainer's sum(((out - gt) W)^2) / batch (slam/network/lstm_model.py:52-55)
and the PoseNet loss sum((x^ - x)^2) + 1100 sum((q^ - q)^2)
(slam/main/vgg16_posenet.py:16-25, slam/network/google_net.py:242-250).
Each returns (loss [1], d loss / d out) as GPU tensors."""
import torch

from . import encoder_ops as ops


def lstm_loss(out, gt, loss_weight):
    loss = torch.empty(1, dtype=torch.float32, device=out.device)
    d_out = torch.empty_like(out)
    ops.weighted_sq_loss(out, gt, loss_weight, 1.0 / out.shape[0], loss, d_out)
    return loss, d_out


def posenet_loss(out7, gt7, beta=1100.0):
    loss = torch.empty(1, dtype=torch.float32, device=out7.device)
    d_out = torch.empty_like(out7)
    ops.posenet_loss(out7, gt7, loss, d_out, beta=beta)
    return loss, d_out


def posenet_multi_head_loss(heads, gt7, weights=(0.3, 0.3, 1.0), beta=1100.0):
    """google_net.py:223 - the three GoogLeNet heads weighted 0.3 / 0.3 / 1."""
    total, grads = None, []
    for out7, w in zip(heads, weights):
        loss, d = posenet_loss(out7, gt7, beta)
        total = loss * w if total is None else total + loss * w
        grads.append(d * w)
    return total, grads
