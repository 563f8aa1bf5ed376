"""Input preparation on the GPU: raw RGB + depth frames -> mean-subtracted
float32 (and bfloat16) [B, 224, 224, 4] batches, the work of
SimpleInputProvider.get_rgbd_file and the per-batch mean subtraction
(slam/network/model_input.py:427-450, 360-362, 517-522) in one C call."""
import torch

from .. import capi


def rgbd_batch(rgb, depth, out_hw=(224, 224), out=None, out_bf16=None):
    """rgb uint8 [B,H,W,3], depth uint16 [B,H,W] (GPU tensors) -> (float32 [B,Ho,Wo,4], float64 mean [4])."""
    if not (rgb.is_cuda and depth.is_cuda and rgb.dtype == torch.uint8 and depth.dtype == torch.uint16):
        raise ValueError("rgb must be a uint8 and depth a uint16 GPU tensor")
    rgb, depth = rgb.contiguous(), depth.contiguous()
    B, H, W, C = rgb.shape
    if C != 3 or depth.shape != (B, H, W):
        raise ValueError("rgb [B,H,W,3] and depth [B,H,W] expected, got %s %s" % (tuple(rgb.shape), tuple(depth.shape)))
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty(B, Ho, Wo, 4, dtype=torch.float32, device=rgb.device)
    elif out.shape != (B, Ho, Wo, 4) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 [B,Ho,Wo,4] tensor")
    if out_bf16 is not None and (out_bf16.shape != out.shape or out_bf16.dtype != torch.bfloat16
                                 or not out_bf16.is_contiguous()):
        raise ValueError("out_bf16 must be a contiguous bfloat16 [B,Ho,Wo,4] tensor")
    lib = capi.load()
    nbytes = lib.slamhip_prep_rgbd_workspace_bytes(B)
    ws = torch.empty(max(nbytes, 8), dtype=torch.uint8, device=rgb.device)
    mean = torch.empty(4, dtype=torch.float64, device=rgb.device)
    capi.check(lib.slamhip_prep_rgbd(capi.ptr(rgb), capi.ptr(depth), B, H, W, Ho, Wo, capi.ptr(out), capi.ptr(out_bf16),
                                     capi.ptr(mean), capi.ptr(ws), ws.numel(), capi.current_stream()),
               "slamhip_prep_rgbd")
    return out, mean


def rgbd_batch_tf(rgb, depth, out_hw=(224, 224), out=None):
    """The reference's in-graph reader (slam/network/model_input.py:185-217) for a batch of
    raw frames: 8-bit depth (upper byte), TensorFlow 0.x bilinear resize, depth * 255 /
    max per frame, no mean subtraction.  -> float32 [B,Ho,Wo,4]."""
    if not (rgb.is_cuda and depth.is_cuda and rgb.dtype == torch.uint8 and depth.dtype == torch.uint16):
        raise ValueError("rgb must be a uint8 and depth a uint16 GPU tensor")
    rgb, depth = rgb.contiguous(), depth.contiguous()
    B, H, W, C = rgb.shape
    if C != 3 or depth.shape != (B, H, W):
        raise ValueError("rgb [B,H,W,3] and depth [B,H,W] expected, got %s %s" % (tuple(rgb.shape), tuple(depth.shape)))
    Ho, Wo = out_hw
    if out is None:
        out = torch.empty(B, Ho, Wo, 4, dtype=torch.float32, device=rgb.device)
    elif out.shape != (B, Ho, Wo, 4) or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 [B,Ho,Wo,4] tensor")
    lib = capi.load()
    ws = torch.empty(max(lib.slamhip_prep_rgbd_workspace_bytes(B), 8), dtype=torch.uint8, device=rgb.device)
    capi.check(lib.slamhip_prep_rgbd_tf(capi.ptr(rgb), capi.ptr(depth), B, H, W, Ho, Wo, capi.ptr(out), capi.ptr(ws),
                                        ws.numel(), capi.current_stream()), "slamhip_prep_rgbd_tf")
    return out
