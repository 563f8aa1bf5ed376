"""The frame-pair path (back-projection, rigid transform, re-projection, masks, bilinear warp, residual sums) written with
plain torch tensor operations on the GPU - float64 geometry, float32 interpolation, every intermediate a tensor in HBM - as a
yardstick for the fused kernel on the bench's workload (64 pairs of 640x480).  Timing only (its sums are compared loosely with
the fused kernel's); not part of the product, not used by any test."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from slam_amd import frame_pairs as fp
from slam_amd.utils import synthetic as syn


def torch_pair_loss(d1, c1, d2, c2, rel, K, factor):
    B, H, W = d1.shape
    fx, fy, cx, cy = K
    dev = d1.device
    z = d1.to(torch.float64) / factor
    col = torch.arange(W, device=dev, dtype=torch.float64).view(1, 1, W)
    row = torch.arange(H, device=dev, dtype=torch.float64).view(1, H, 1)
    x = (col - cx) * z / fx
    y = (row - cy) * z / fy
    valid0 = (x != 0) & (y != 0) & (z != 0)
    R, t = rel[:, 0, :3, :3], rel[:, 0, :3, 3]
    p = torch.stack([x, y, z], -1).reshape(B, H * W, 3)
    q = torch.bmm(p, R) + t.view(B, 1, 3)                    # the reference's row-vector convention
    x2, y2, z2 = q[..., 0].reshape(B, H, W), q[..., 1].reshape(B, H, W), q[..., 2].reshape(B, H, W)
    u = x2 * fx / z2 + cx
    v = y2 * fy / z2 + cy
    inb = valid0 & (u >= 0) & (u <= W) & (v >= 0) & (v <= H)
    wv = valid0 & (u >= 0) & (u <= W - 1) & (v >= 0) & (v <= H - 1) & (z2 > 0)
    u = torch.where(wv, u, torch.zeros_like(u))
    v = torch.where(wv, v, torch.zeros_like(v))
    x0, y0 = u.floor().long(), v.floor().long()
    x1, y1 = (x0 + 1).clamp(max=W - 1), (y0 + 1).clamp(max=H - 1)
    fu, fv = (u - u.floor()).float(), (v - v.floor()).float()
    bi = torch.arange(B, device=dev).view(B, 1, 1)
    c2f, d2f = c2[:, 0].float(), d2[:, 0].float()

    def bil(img):
        a, b_, c_, e = img[bi, y0, x0], img[bi, y0, x1], img[bi, y1, x0], img[bi, y1, x1]
        if img.dim() == 4:
            fuu, fvv = fu.unsqueeze(-1), fv.unsqueeze(-1)
        else:
            fuu, fvv = fu, fv
        top, bot = a + fuu * (b_ - a), c_ + fuu * (e - c_)
        return top + fvv * (bot - top)
    w_rgb = bil(c2f)
    w_d = bil(d2f) / factor
    d2r = d2[:, 0]
    taps = (d2r[bi, y0, x0] != 0) & (d2r[bi, y0, x1] != 0) & (d2r[bi, y1, x0] != 0) & (d2r[bi, y1, x1] != 0)
    gv = wv & taps
    rp = torch.where(wv.unsqueeze(-1), w_rgb - c1.float(), torch.zeros_like(w_rgb)).double()
    rg = torch.where(gv, (w_d.double() - z2).float(), torch.zeros_like(w_d)).double()
    sums = torch.stack([rp.abs().sum((1, 2, 3)), (rp * rp).sum((1, 2, 3)), rg.abs().sum((1, 2)), (rg * rg).sum((1, 2))], -1)
    counts = torch.stack([valid0.sum((1, 2)), inb.sum((1, 2)), wv.sum((1, 2)), gv.sum((1, 2))], -1)
    return sums, counts


def main():
    B, H, W = 64, 480, 640
    K = fp.intrinsics(1)
    gt = torch.from_numpy(syn.pose_pairs(B, seed=3)).cuda()
    rel = fp.relative_pose(gt[:, 0].contiguous(), gt[:, 1].contiguous()).reshape(B, 1, 4, 4)
    d1 = torch.from_numpy(syn.depth_frames(B, H, W, seed=1, kind="plane").astype(np.int32)).cuda()
    c1 = torch.from_numpy(syn.rgb_frames(B, H, W, seed=1)).cuda()
    d2 = torch.from_numpy(syn.depth_frames(B, H, W, seed=2, kind="plane").astype(np.int32)).cuda().reshape(B, 1, H, W)
    c2 = torch.from_numpy(syn.rgb_frames(B, H, W, seed=2)).cuda().reshape(B, 1, H, W, 3)
    for _ in range(3):
        s, c = torch_pair_loss(d1, c1, d2, c2, rel, K, 1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        s, c = torch_pair_loss(d1, c1, d2, c2, rel, K, 1.0)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    fs, fc = fp.pair_loss(d1.to(torch.uint16), c1, d2.to(torch.uint16), c2, rel, K, 1.0)
    same_counts = bool((fc.reshape(B, 4).to(torch.int64) & 0xFFFFFFFF == c).all())
    rel_err = float(((fs.reshape(B, 4) - s).abs() / (s.abs() + 1e-9)).max())
    print("plain torch on the GPU: %.2f ms per %d pairs = %.0f pairs/s (counts equal the fused kernel's: %s, sums within %.1e)"
          % (ms, B, B / ms * 1e3, same_counts, rel_err))


if __name__ == "__main__":
    main()
