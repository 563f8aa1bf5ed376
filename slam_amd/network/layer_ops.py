"""Typed wrappers of the layer entry points of libslamhip.so that the GoogLeNet
encoder uses (csrc/layers.hip): im2col / col2im, slice epilogues, general pooling,
local response normalisation.  GPU torch tensors in, launches on the current
stream, shapes checked on the host."""
import ctypes

import torch

from .. import capi
from .encoder_ops import BF16, F32, _call, _chk


def conv_out_shape(H, W, KH, KW, stride, same=True):
    ho, wo = ctypes.c_int(), ctypes.c_int()
    capi.check(capi.load().slamhip_conv_out_shape(H, W, KH, KW, stride, int(same), ctypes.addressof(ho),
                                                  ctypes.addressof(wo)), "slamhip_conv_out_shape")
    return ho.value, wo.value


def padded_k(KH, KW, C):
    return (KH * KW * C + 63) // 64 * 64


def im2col(x, col, KH, KW, stride):
    B, H, W, C = _chk(x, BF16, "x").shape
    ho, wo = conv_out_shape(H, W, KH, KW, stride)
    assert C % 8 == 0 and _chk(col, BF16, "col").shape == (B * ho * wo, padded_k(KH, KW, C))
    _call("slamhip_im2col_bf16", capi.ptr(x), capi.ptr(col), B, H, W, C, KH, KW, stride)


def col2im_add(dcol, dx, KH, KW, stride):
    B, H, W, C = _chk(dx, BF16, "dx").shape
    ho, wo = conv_out_shape(H, W, KH, KW, stride)
    assert C % 8 == 0 and _chk(dcol, BF16, "dcol").shape == (B * ho * wo, padded_k(KH, KW, C))
    _call("slamhip_col2im_add_bf16", capi.ptr(dcol), capi.ptr(dx), B, H, W, C, KH, KW, stride)


def _slice_ok(buf, M, N, ld, off):
    assert buf.is_cuda and buf.is_contiguous() and buf.numel() == M * ld and 0 <= off and off + N <= ld


def bias_act_slice(z, bias, N, y_bf16=None, y_f32=None, ld=None, off=0, relu=True):
    """y[m, off:off+N] = act(z[m, :N] + bias[:N]); y is a [M, ld] buffer (any leading shape)."""
    M, Np = _chk(z, F32, "z").shape
    ld = N if ld is None else ld
    assert _chk(bias, F32, "bias").numel() >= N and N <= Np
    for y, dt in ((y_bf16, BF16), (y_f32, F32)):
        if y is not None:
            assert y.dtype == dt
            _slice_ok(y, M, N, ld, off)
    _call("slamhip_bias_act_slice_fwd", capi.ptr(z), capi.ptr(bias), capi.ptr(y_bf16), capi.ptr(y_f32), M, N, Np, ld,
          off, int(relu))


def bias_act_slice_bwd(dy, y_bf16, dz, dbias, N, ld=None, off=0, relu=True):
    """dz[m, :N] = dy[m, off:off+N] gated by y > 0; dz[m, N:] = 0; dbias = column sums."""
    M, Np = _chk(dz, BF16, "dz").shape
    ld = N if ld is None else ld
    assert dy.dtype in (BF16, F32) and _chk(dbias, F32, "dbias").numel() == Np
    _slice_ok(dy, M, N, ld, off)
    if relu:
        assert y_bf16.dtype == BF16
        _slice_ok(y_bf16, M, N, ld, off)
    _call("slamhip_bias_act_slice_bwd", capi.ptr(dy) if dy.dtype == BF16 else None,
          capi.ptr(dy) if dy.dtype == F32 else None, capi.ptr(y_bf16) if relu else None, capi.ptr(dz), capi.ptr(dbias),
          M, N, Np, ld, off, int(relu))


def pool(x, y, K, stride, same=True, is_max=True):
    B, H, W, C = _chk(x, BF16, "x").shape
    ho, wo = conv_out_shape(H, W, K, K, stride, same)
    assert _chk(y, BF16, "y").shape == (B, ho, wo, C)
    _call("slamhip_pool_fwd", capi.ptr(x), capi.ptr(y), B, H, W, C, K, stride, int(same), int(is_max))


def pool_bwd_add(x, dy, dx, K, stride, same=True, is_max=True):
    B, H, W, C = _chk(x, BF16, "x").shape
    ho, wo = conv_out_shape(H, W, K, K, stride, same)
    assert _chk(dy, BF16, "dy").shape == (B, ho, wo, C) and _chk(dx, BF16, "dx").shape == x.shape
    _call("slamhip_pool_bwd_add", capi.ptr(x), capi.ptr(dy), capi.ptr(dx), B, H, W, C, K, stride, int(same), int(is_max))


def lrn(x, y, radius, alpha, beta, bias=1.0):
    C = _chk(x, BF16, "x").shape[-1]
    assert _chk(y, BF16, "y").shape == x.shape
    _call("slamhip_lrn_fwd", capi.ptr(x), capi.ptr(y), x.numel() // C, C, radius, float(bias), float(alpha), float(beta))


def lrn_bwd_add(x, dy, dx, radius, alpha, beta, bias=1.0):
    C = _chk(x, BF16, "x").shape[-1]
    assert _chk(dy, BF16, "dy").shape == x.shape and _chk(dx, BF16, "dx").shape == x.shape
    _call("slamhip_lrn_bwd_add", capi.ptr(x), capi.ptr(dy), capi.ptr(dx), x.numel() // C, C, radius, float(bias),
          float(alpha), float(beta))


def pad_cast(src, dst):
    C, Cp = _chk(src, F32, "src").shape[-1], _chk(dst, BF16, "dst").shape[-1]
    assert Cp >= C and src.numel() // C == dst.numel() // Cp
    _call("slamhip_pad_cast_bf16", capi.ptr(src), capi.ptr(dst), src.numel() // C, C, Cp)
