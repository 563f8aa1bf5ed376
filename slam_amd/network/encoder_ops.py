"""Typed wrappers of the encoder entry points of libslamhip.so.

Every function takes torch tensors that already live on the GPU (torch owns the
memory and the stream, nothing else), checks shapes/dtypes on the host so a
wrong call cannot reach a kernel, and launches on the current stream.
Activations are NHWC bfloat16; parameters float32 in the reference's layouts
(slam/network/cnn_model.py:90-95, 120-121).
"""
import torch

from .. import capi

BF16 = torch.bfloat16
F32 = torch.float32


def _chk(t, dtype, what):
    if t.dtype != dtype or not t.is_contiguous() or not t.is_cuda:
        raise ValueError("%s: expected a contiguous %s GPU tensor, got %s %s contiguous=%s" %
                         (what, dtype, t.dtype, t.device, t.is_contiguous()))
    return t


def _call(name, *args):
    capi.check(getattr(capi.load(), name)(*args, capi.current_stream()), name)


def zero_fill(t):
    """Clears a contiguous GPU tensor with the library's kernel (no memset node under graph capture)."""
    assert t.is_cuda and t.is_contiguous()
    _call("slamhip_zero_fill", capi.ptr(t), t.numel() * t.element_size())


def caller_clears_accumulators(on):
    """The calling thread clears sums / red / dw accumulators itself; returns the previous setting."""
    return bool(capi.load().slamhip_caller_clears_accumulators(int(bool(on))))


def deterministic():
    """True when SLAMHIP_DETERMINISTIC=1 selected fixed-order sums for this process."""
    return bool(capi.load().slamhip_deterministic())


def conv_weight_layouts(w_hwio, w_fwd=None, w_dgrad=None):
    """[3,3,Cin,Cout] float32 -> [Cout, 9*Cin] and/or [Cin, 9*Cout] bfloat16 operands."""
    _chk(w_hwio, F32, "w_hwio")
    kh, kw, cin, cout = w_hwio.shape
    assert (kh, kw) == (3, 3)
    if w_fwd is not None:
        assert _chk(w_fwd, BF16, "w_fwd").shape == (cout, 9 * cin)
    if w_dgrad is not None:
        assert _chk(w_dgrad, BF16, "w_dgrad").shape == (cin, 9 * cout)
    _call("slamhip_conv_weight_layouts", capi.ptr(w_hwio), capi.ptr(w_fwd), capi.ptr(w_dgrad), cin, cout)


def conv_weight_layouts_multi(flat_params, layers):
    """conv_weight_layouts for several filters of one flat float32 buffer in one launch.
    layers: [(offset in floats, Cin, Cout, w_fwd or None, w_dgrad or None), ...], at most 16."""
    import ctypes
    _chk(flat_params, F32, "flat_params")
    n = len(layers)
    assert 0 < n <= 16
    for off, cin, cout, wf, wd in layers:
        assert 0 <= off and off + 9 * cin * cout <= flat_params.numel()
        if wf is not None:
            assert _chk(wf, BF16, "w_fwd").shape == (cout, 9 * cin)
        if wd is not None:
            assert _chk(wd, BF16, "w_dgrad").shape == (cin, 9 * cout)
    offs = (ctypes.c_longlong * n)(*[l[0] for l in layers])
    cins = (ctypes.c_int * n)(*[l[1] for l in layers])
    couts = (ctypes.c_int * n)(*[l[2] for l in layers])
    wf = (ctypes.c_void_p * n)(*[capi.ptr(l[3]) for l in layers])
    wd = (ctypes.c_void_p * n)(*[capi.ptr(l[4]) for l in layers])
    _call("slamhip_conv_weight_layouts_multi", capi.ptr(flat_params), n, offs, cins, couts, wf, wd)


def conv3x3(act, w_nk, out_bf16=None, out_f32=None, flip=False):
    """3x3 SAME convolution (flip=False) or its input gradient (flip=True)."""
    _chk(act, BF16, "act")
    B, H, W, C = act.shape
    N = _chk(w_nk, BF16, "w_nk").shape[0]
    assert w_nk.shape == (N, 9 * C) and C % 64 == 0
    for o, dt in ((out_bf16, BF16), (out_f32, F32)):
        if o is not None:
            assert _chk(o, dt, "out").shape == (B, H, W, N)
    _call("slamhip_conv3x3_igemm_bf16", capi.ptr(act), capi.ptr(w_nk), capi.ptr(out_bf16), capi.ptr(out_f32),
          B, H, W, C, N, int(flip))


def gemm_nt(a, b, out_bf16=None, out_f32=None):
    """out[m, n] = sum_k a[m, k] * b[n, k]."""
    M, K = _chk(a, BF16, "a").shape
    N = _chk(b, BF16, "b").shape[0]
    assert b.shape == (N, K) and K % 64 == 0
    for o, dt in ((out_bf16, BF16), (out_f32, F32)):
        if o is not None:
            assert _chk(o, dt, "out").shape == (M, N)
    _call("slamhip_gemm_nt_bf16", capi.ptr(a), capi.ptr(b), capi.ptr(out_bf16), capi.ptr(out_f32), M, N, K)


def wgrad(x, dy, dw):
    """Filter gradient [3,3,Cin,Cout] (4-d x) or x^T dy [Cin,Cout] (2-d x), float32."""
    _chk(x, BF16, "x"), _chk(dy, BF16, "dy"), _chk(dw, F32, "dw")
    if x.dim() == 4:
        B, H, W, cin = x.shape
        cout = dy.shape[3]
        assert dy.shape == (B, H, W, cout) and dw.shape == (3, 3, cin, cout)
        taps = 9
    else:
        B, cin = x.shape
        cout = dy.shape[1]
        assert dy.shape == (B, cout) and dw.shape == (cin, cout)
        H = W = taps = 1
    assert cin % 64 == 0 and cout % 64 == 0
    _call("slamhip_wgrad_bf16", capi.ptr(x), capi.ptr(dy), capi.ptr(dw), B, H, W, cin, cout, taps)


def conv3x3_small(x, w_hwio, out):
    B, H, W, cin = _chk(x, BF16, "x").shape
    cout = _chk(w_hwio, F32, "w").shape[3]
    assert w_hwio.shape == (3, 3, cin, cout) and _chk(out, BF16, "out").shape == (B, H, W, cout)
    _call("slamhip_conv3x3_small_fwd", capi.ptr(x), capi.ptr(w_hwio), capi.ptr(out), B, H, W, cin, cout)


def conv3x3_small_wgrad(x, dy, dw):
    B, H, W, cin = _chk(x, BF16, "x").shape
    cout = _chk(dy, BF16, "dy").shape[3]
    assert dy.shape == (B, H, W, cout) and _chk(dw, F32, "dw").shape == (3, 3, cin, cout) and 256 % cout == 0
    _call("slamhip_conv3x3_small_wgrad", capi.ptr(x), capi.ptr(dy), capi.ptr(dw), B, H, W, cin, cout)


def channel_stats(z, sums2c):
    C = _chk(z, BF16, "z").shape[-1]
    P = z.numel() // C
    assert _chk(sums2c, F32, "sums").shape == (2, C) and C % 64 == 0
    _call("slamhip_channel_stats", capi.ptr(z), capi.ptr(sums2c), P, C)


def bn_relu(z, sums2c, gamma, beta, y, eps):
    C = _chk(z, BF16, "z").shape[-1]
    P = z.numel() // C
    assert _chk(y, BF16, "y").shape == z.shape and sums2c.shape == (2, C)
    assert _chk(gamma, F32, "gamma").shape == (C,) and _chk(beta, F32, "beta").shape == (C,)
    _call("slamhip_bn_relu_fwd", capi.ptr(z), capi.ptr(sums2c), capi.ptr(gamma), capi.ptr(beta), capi.ptr(y), P, C,
          float(eps))


def bn_relu_bwd(dy, z, sums2c, gamma, beta, red2c, dz, eps, phases=3):
    """dz, and red2c[0] = dbeta, red2c[1] = dgamma.  The ReLU gate is recomputed from z.
    phases: 1 = the reduction pass alone (red2c), 2 = the pass that writes dz from red2c, 3 = both
    (statistics kept over several ranks: red2c is averaged over the ranks between the two)."""
    C = _chk(z, BF16, "z").shape[-1]
    P = z.numel() // C
    for t in (dy, dz):
        assert _chk(t, BF16, "bn tensor").shape == z.shape
    assert _chk(red2c, F32, "red").shape == (2, C) and sums2c.shape == (2, C)
    assert _chk(gamma, F32, "gamma").shape == (C,) and _chk(beta, F32, "beta").shape == (C,)
    _call("slamhip_bn_relu_bwd_phases", capi.ptr(dy), capi.ptr(z), capi.ptr(sums2c), capi.ptr(gamma), capi.ptr(beta),
          capi.ptr(red2c), capi.ptr(dz), P, C, float(eps), int(phases))


def bn_relu_bwd_first_layer(dy, z, sums2c, gamma, beta, red2c, x, dw, eps):
    """bn_relu_bwd and conv3x3_small_wgrad of the first layer (4 -> 64 channels) without the
    dz tensor in between: red2c [2,64] and dw [3,3,4,64] come out."""
    B, H, W, C = _chk(z, BF16, "z").shape
    assert C == 64 and _chk(dy, BF16, "dy").shape == z.shape and _chk(x, BF16, "x").shape == (B, H, W, 4)
    assert _chk(sums2c, F32, "sums").shape == (2, C) and _chk(red2c, F32, "red").shape == (2, C)
    assert _chk(gamma, F32, "gamma").shape == (C,) and _chk(beta, F32, "beta").shape == (C,)
    assert _chk(dw, F32, "dw").shape == (3, 3, 4, C)
    _call("slamhip_bn_relu_bwd_first_layer", capi.ptr(dy), capi.ptr(z), capi.ptr(sums2c), capi.ptr(gamma),
          capi.ptr(beta), capi.ptr(red2c), capi.ptr(x), capi.ptr(dw), B, H, W, float(eps))


def conv3x3_c4_bn_relu(x, w_hwio, gamma, beta, y, sums2c, workspace, eps):
    """First layer (4 -> 64 channels) forward without a stored z: statistics pass, then y = relu(bn(conv(x)))."""
    B, H, W, cin = _chk(x, BF16, "x").shape
    assert cin == 4 and _chk(w_hwio, F32, "w").shape == (3, 3, 4, 64) and _chk(y, BF16, "y").shape == (B, H, W, 64)
    assert _chk(sums2c, F32, "sums").shape == (2, 64) and workspace.is_cuda and workspace.is_contiguous()
    assert _chk(gamma, F32, "gamma").shape == (64,) and _chk(beta, F32, "beta").shape == (64,)
    _call("slamhip_conv3x3_c4_bn_relu_fwd", capi.ptr(x), capi.ptr(w_hwio), capi.ptr(gamma), capi.ptr(beta), capi.ptr(y),
          capi.ptr(sums2c), capi.ptr(workspace), workspace.numel() * workspace.element_size(), B, H, W, float(eps))


def bn_relu_bwd_first_layer_recompute(dy, x, w_hwio, sums2c, gamma, beta, red2c, dw, workspace, eps):
    """bn_relu_bwd_first_layer with z recomputed from x and the filter instead of read."""
    B, H, W, cin = _chk(x, BF16, "x").shape
    assert cin == 4 and _chk(dy, BF16, "dy").shape == (B, H, W, 64) and _chk(w_hwio, F32, "w").shape == (3, 3, 4, 64)
    assert _chk(sums2c, F32, "sums").shape == (2, 64) and _chk(red2c, F32, "red").shape == (2, 64)
    assert _chk(gamma, F32, "gamma").shape == (64,) and _chk(beta, F32, "beta").shape == (64,)
    assert _chk(dw, F32, "dw").shape == (3, 3, 4, 64) and workspace.is_cuda and workspace.is_contiguous()
    _call("slamhip_bn_relu_bwd_first_layer_recompute", capi.ptr(dy), capi.ptr(x), capi.ptr(w_hwio), capi.ptr(sums2c),
          capi.ptr(gamma), capi.ptr(beta), capi.ptr(red2c), capi.ptr(dw), capi.ptr(workspace),
          workspace.numel() * workspace.element_size(), B, H, W, float(eps))


def bn_relu_pool(z, sums2c, gamma, beta, y, pooled, eps):
    """bn_relu and the 2x2 max-pool after it in one pass: y (like z; None = not written) and
    pooled [B,H/2,W/2,C]."""
    B, H, W, C = _chk(z, BF16, "z").shape
    assert H % 2 == 0 and W % 2 == 0 and C % 8 == 0
    assert y is None or _chk(y, BF16, "y").shape == z.shape
    assert _chk(pooled, BF16, "pooled").shape == (B, H // 2, W // 2, C)
    assert _chk(sums2c, F32, "sums").shape == (2, C)
    assert _chk(gamma, F32, "gamma").shape == (C,) and _chk(beta, F32, "beta").shape == (C,)
    _call("slamhip_bn_relu_pool_fwd", capi.ptr(z), capi.ptr(sums2c), capi.ptr(gamma), capi.ptr(beta), capi.ptr(y),
          capi.ptr(pooled), B, H, W, C, float(eps))


def bn_relu_pool_bwd(dpooled, z, sums2c, gamma, beta, red2c, dz, eps, phases=3):
    """Max-pool backward, ReLU gate and batch-norm backward in one step: the pooled gradient
    [B,H/2,W/2,C] -> dz (like z); red2c and phases as bn_relu_bwd."""
    B, H, W, C = _chk(z, BF16, "z").shape
    assert H % 2 == 0 and W % 2 == 0 and C % 64 == 0
    assert _chk(dpooled, BF16, "dpooled").shape == (B, H // 2, W // 2, C) and _chk(dz, BF16, "dz").shape == z.shape
    assert _chk(red2c, F32, "red").shape == (2, C) and _chk(sums2c, F32, "sums").shape == (2, C)
    assert _chk(gamma, F32, "gamma").shape == (C,) and _chk(beta, F32, "beta").shape == (C,)
    _call("slamhip_bn_relu_pool_bwd_phases", capi.ptr(dpooled), capi.ptr(z), capi.ptr(sums2c), capi.ptr(gamma),
          capi.ptr(beta), capi.ptr(red2c), capi.ptr(dz), B, H, W, C, float(eps), int(phases))


def bias_act(z, bias, y_bf16=None, y_f32=None, relu=True):
    M, N = _chk(z, F32, "z").shape
    assert _chk(bias, F32, "bias").shape == (N,)
    for o, dt in ((y_bf16, BF16), (y_f32, F32)):
        if o is not None:
            assert _chk(o, dt, "y").shape == (M, N)
    _call("slamhip_bias_act_fwd", capi.ptr(z), capi.ptr(bias), capi.ptr(y_bf16), capi.ptr(y_f32), M, N, int(relu))


def bias_act_bwd(dy, y_bf16, dz, dbias, relu=True):
    M, N = _chk(dy, F32, "dy").shape
    assert _chk(dz, BF16, "dz").shape == (M, N) and _chk(dbias, F32, "dbias").shape == (N,)
    if relu:
        assert _chk(y_bf16, BF16, "y").shape == (M, N)
    _call("slamhip_bias_act_bwd", capi.ptr(dy), capi.ptr(y_bf16) if relu else None, capi.ptr(dz), capi.ptr(dbias),
          M, N, int(relu))


def maxpool2(x, y):
    B, H, W, C = _chk(x, BF16, "x").shape
    assert _chk(y, BF16, "y").shape == (B, H // 2, W // 2, C) and C % 8 == 0
    _call("slamhip_maxpool2_fwd", capi.ptr(x), capi.ptr(y), B, H, W, C)


def maxpool2_bwd(x, y, dy, dx):
    B, H, W, C = _chk(x, BF16, "x").shape
    assert _chk(y, BF16, "y").shape == (B, H // 2, W // 2, C) and _chk(dy, BF16, "dy").shape == y.shape
    assert _chk(dx, BF16, "dx").shape == x.shape
    _call("slamhip_maxpool2_bwd", capi.ptr(x), capi.ptr(y), capi.ptr(dy), capi.ptr(dx), B, H, W, C)


def cast_bf16(src, dst):
    assert _chk(src, F32, "src").numel() == _chk(dst, BF16, "dst").numel()
    _call("slamhip_cast_f32_bf16", capi.ptr(src), capi.ptr(dst), src.numel())


def transpose_cast(src, dst):
    R, C = _chk(src, F32, "src").shape
    assert _chk(dst, BF16, "dst").shape == (C, R)
    _call("slamhip_transpose_cast_bf16", capi.ptr(src), capi.ptr(dst), R, C)


def transpose_bf16(src, dst):
    """bf16 [R][C] -> bf16 [C][R]."""
    R, C = _chk(src, BF16, "src").shape
    assert _chk(dst, BF16, "dst").shape == (C, R)
    _call("slamhip_transpose_bf16", capi.ptr(src), capi.ptr(dst), R, C)


def l2loss(out, gt, loss_weight, loss, d_out=None):
    B, D = _chk(out, F32, "out").shape
    assert _chk(gt, F32, "gt").shape == (B, D) and _chk(loss_weight, F32, "loss_weight").shape == (D, D)
    assert _chk(loss, F32, "loss").numel() == 1 and D <= 8
    if d_out is not None:
        assert _chk(d_out, F32, "d_out").shape == (B, D)
    _call("slamhip_l2loss", capi.ptr(out), capi.ptr(gt), capi.ptr(loss_weight), capi.ptr(loss), capi.ptr(d_out), B, D)


def adam_step(param, grad, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, shadow=None):
    """One Adam update of the flat parameter buffer.  `shadow` (bf16, at most as long as
    `param`) receives the updated leading parameters as bf16 in the same pass."""
    n = _chk(param, F32, "param").numel()
    for t in (grad, m, v):
        assert _chk(t, F32, "adam tensor").numel() == n
    if shadow is None:
        _call("slamhip_adam_step", capi.ptr(param), capi.ptr(grad), capi.ptr(m), capi.ptr(v), n, float(lr),
              float(beta1), float(beta2), float(eps), int(step), float(grad_scale))
    else:
        assert _chk(shadow, BF16, "shadow").numel() <= n
        _call("slamhip_adam_step_shadow", capi.ptr(param), capi.ptr(grad), capi.ptr(m), capi.ptr(v), n, float(lr),
              float(beta1), float(beta2), float(eps), int(step), float(grad_scale), capi.ptr(shadow), shadow.numel())


def adam_lr_t(lr, step, beta1=0.9, beta2=0.999):
    """The bias-corrected step size of Adam's update number `step` (>= 1), in float32 as the library computes it."""
    import numpy as np
    f = np.float32
    return float(f(lr) * np.sqrt(f(1) - np.power(f(beta2), f(step))) / (f(1) - np.power(f(beta1), f(step))))


def adam_step_dev(param, grad, m, v, lr_t_dev, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0, shadow=None):
    """adam_step with the bias-corrected step size read from one float of device memory (`lr_t_dev`):
    the form a training step recorded in a HIP graph uses."""
    n = _chk(param, F32, "param").numel()
    for t in (grad, m, v):
        assert _chk(t, F32, "adam tensor").numel() == n
    assert _chk(lr_t_dev, F32, "lr_t_dev").numel() >= 1
    _call("slamhip_adam_step_dev", capi.ptr(param), capi.ptr(grad), capi.ptr(m), capi.ptr(v), n, capi.ptr(lr_t_dev),
          float(beta1), float(beta2), float(eps), float(grad_scale), capi.ptr(shadow),
          0 if shadow is None else shadow.numel())


def sgd_step(param, grad, lr, grad_scale=1.0):
    n = _chk(param, F32, "param").numel()
    assert _chk(grad, F32, "grad").numel() == n
    _call("slamhip_sgd_step", capi.ptr(param), capi.ptr(grad), n, float(lr), float(grad_scale))


def rmsprop_step(param, grad, ms, lr, decay=0.9, eps=1e-10, grad_scale=1.0):
    n = _chk(param, F32, "param").numel()
    assert _chk(grad, F32, "grad").numel() == n and _chk(ms, F32, "ms").numel() == n
    _call("slamhip_rmsprop_step", capi.ptr(param), capi.ptr(grad), capi.ptr(ms), n, float(lr), float(decay),
          float(eps), float(grad_scale))


def weighted_sq_loss(out, gt, loss_weight, scale, loss, d_out=None):
    """loss = scale * sum(((out - gt) W)^2); scale 1/B is the LSTM trainer's loss (lstm_model.py:52-55)."""
    B, D = _chk(out, F32, "out").shape
    assert _chk(gt, F32, "gt").shape == (B, D) and _chk(loss_weight, F32, "loss_weight").shape == (D, D)
    assert _chk(loss, F32, "loss").numel() == 1 and D <= 8
    if d_out is not None:
        assert _chk(d_out, F32, "d_out").shape == (B, D)
    _call("slamhip_weighted_sq_loss", capi.ptr(out), capi.ptr(gt), capi.ptr(loss_weight), float(scale), capi.ptr(loss),
          capi.ptr(d_out), B, D)


def posenet_loss(out7, gt7, loss, d_out=None, beta=1100.0):
    """vgg16_posenet.py:16-25."""
    B, D = _chk(out7, F32, "out").shape
    assert D == 7 and _chk(gt7, F32, "gt").shape == (B, 7) and _chk(loss, F32, "loss").numel() == 1
    if d_out is not None:
        assert _chk(d_out, F32, "d_out").shape == (B, 7)
    _call("slamhip_posenet_loss", capi.ptr(out7), capi.ptr(gt7), float(beta), capi.ptr(loss), capi.ptr(d_out), B)


def bias_relu(z, bias, y):
    C = _chk(z, BF16, "z").shape[-1]
    assert _chk(y, BF16, "y").shape == z.shape and _chk(bias, F32, "bias").shape == (C,) and C % 8 == 0
    _call("slamhip_bias_relu_fwd", capi.ptr(z), capi.ptr(bias), capi.ptr(y), z.numel() // C, C)


def bias_relu_bwd(dy, y, dz, dbias):
    C = _chk(y, BF16, "y").shape[-1]
    assert _chk(dy, BF16, "dy").shape == y.shape and _chk(dz, BF16, "dz").shape == y.shape
    assert _chk(dbias, F32, "dbias").shape == (C,) and C % 64 == 0
    _call("slamhip_bias_relu_bwd", capi.ptr(dy), capi.ptr(y), capi.ptr(dz), capi.ptr(dbias), y.numel() // C, C)


def conv_stats_workspace_bytes(pixels, N):
    return int(capi.load().slamhip_conv_stats_workspace_bytes(int(pixels), int(N)))


def conv3x3_stats(act, w_nk, out_bf16, sums2c, workspace):
    """Forward convolution + per-channel sums of its output (see channel_stats) in one pass."""
    B, H, W, C = _chk(act, BF16, "act").shape
    N = _chk(w_nk, BF16, "w_nk").shape[0]
    assert w_nk.shape == (N, 9 * C) and C % 64 == 0 and _chk(out_bf16, BF16, "out").shape == (B, H, W, N)
    assert _chk(sums2c, F32, "sums").shape == (2, N) and workspace.is_cuda and workspace.is_contiguous()
    _call("slamhip_conv3x3_igemm_bf16_stats", capi.ptr(act), capi.ptr(w_nk), capi.ptr(out_bf16), B, H, W, C, N,
          capi.ptr(sums2c), capi.ptr(workspace), workspace.numel() * workspace.element_size())


def conv3x3_small_stats(x, w_hwio, out, sums2c, workspace):
    B, H, W, cin = _chk(x, BF16, "x").shape
    cout = _chk(w_hwio, F32, "w").shape[3]
    assert w_hwio.shape == (3, 3, cin, cout) and _chk(out, BF16, "out").shape == (B, H, W, cout)
    assert _chk(sums2c, F32, "sums").shape == (2, cout) and workspace.is_cuda and workspace.is_contiguous()
    _call("slamhip_conv3x3_small_fwd_stats", capi.ptr(x), capi.ptr(w_hwio), capi.ptr(out), B, H, W, cin, cout,
          capi.ptr(sums2c), capi.ptr(workspace), workspace.numel() * workspace.element_size())


def lstm_head_fwd(x, state_in, W, bias, fc_w, fc_b, gates, state_out, out, forget_bias=1.0):
    """Stack of LSTM cells for one time step + linear output layer (float32)."""
    B, n = _chk(x, F32, "x").shape
    L = _chk(W, F32, "W").shape[0]
    D = _chk(fc_w, F32, "fc_w").shape[1]
    assert W.shape == (L, 2 * n, 4 * n) and _chk(bias, F32, "bias").shape == (L, 4 * n)
    assert fc_w.shape == (n, D) and _chk(fc_b, F32, "fc_b").shape == (D,)
    assert _chk(state_in, F32, "state_in").shape == (B, 2 * n * L)
    assert _chk(state_out, F32, "state_out").shape == (B, 2 * n * L)
    assert _chk(gates, F32, "gates").shape == (B, L, 4 * n) and _chk(out, F32, "out").shape == (B, D)
    _call("slamhip_lstm_head_fwd", capi.ptr(x), capi.ptr(state_in), capi.ptr(W), capi.ptr(bias), capi.ptr(fc_w),
          capi.ptr(fc_b), float(forget_bias), B, n, L, D, capi.ptr(gates), capi.ptr(state_out), capi.ptr(out))


def lstm_head_bwd(d_out, x, state_in, W, fc_w, gates, state_out, d_pre, d_x, dW, dbias, dfc_w, dfc_b):
    B, n = _chk(x, F32, "x").shape
    L = _chk(W, F32, "W").shape[0]
    D = _chk(fc_w, F32, "fc_w").shape[1]
    assert W.shape == (L, 2 * n, 4 * n) and fc_w.shape == (n, D)
    assert _chk(d_out, F32, "d_out").shape == (B, D) and _chk(d_x, F32, "d_x").shape == (B, n)
    for t in (state_in, state_out):
        assert _chk(t, F32, "state").shape == (B, 2 * n * L)
    for t in (gates, d_pre):
        assert _chk(t, F32, "gates").shape == (B, L, 4 * n)
    assert _chk(dW, F32, "dW").shape == W.shape and _chk(dbias, F32, "dbias").shape == (L, 4 * n)
    assert _chk(dfc_w, F32, "dfc_w").shape == (n, D) and _chk(dfc_b, F32, "dfc_b").shape == (D,)
    _call("slamhip_lstm_head_bwd", capi.ptr(d_out), capi.ptr(x), capi.ptr(state_in), capi.ptr(W), capi.ptr(fc_w),
          capi.ptr(gates), capi.ptr(state_out), B, n, L, D, capi.ptr(d_pre), capi.ptr(d_x), capi.ptr(dW),
          capi.ptr(dbias), capi.ptr(dfc_w), capi.ptr(dfc_b))
