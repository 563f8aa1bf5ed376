"""GPU parity of the encoder kernels (through the C ABI) against plain PyTorch
float32 on the CPU.  Inputs are rounded to bfloat16 first, so the only
differences are the float32 accumulation order and the final bfloat16 rounding
of outputs (2^-8 relative).  The reference's own numerics (TensorFlow 0.x, its
weights file) are not available: this parity is unpinned by the reference and
anchored on the operator definitions of slam/network/cnn_model.py:100-123.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF = torch.bfloat16


@pytest.fixture(scope="module")
def ops():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slam_amd.network import encoder_ops
    return encoder_ops


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(BF).float()


def close(got, want, rtol=1.5e-2, atol=None):
    got = got.detach().float().cpu()
    want = want.detach().float()
    atol = atol if atol is not None else rtol * float(want.abs().max() + 1e-12)
    err = (got - want).abs()
    assert torch.all(err <= atol + rtol * want.abs()), "max err %g (ref max %g)" % (float(err.max()), float(want.abs().max()))


def conv_ref(x_nhwc, w_hwio):
    return F.conv2d(x_nhwc.permute(0, 3, 1, 2), w_hwio.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,H,W,C,N", [(2, 9, 11, 64, 64), (1, 14, 14, 128, 128), (3, 7, 5, 64, 256), (1, 4, 4, 192, 64)])
def test_conv_fwd_dgrad_wgrad(ops, B, H, W, C, N):
    x = rnd(B, H, W, C, seed=1).requires_grad_()
    w = rnd(3, 3, C, N, seed=2, scale=0.05).requires_grad_()
    y = conv_ref(x, w)
    dy = rnd(B, H, W, N, seed=3)
    y.backward(dy)
    dev = "cuda"
    xd, wd, dyd = x.detach().to(dev).to(BF), w.detach().to(dev), dy.to(dev).to(BF)
    w_fwd = torch.empty(N, 9 * C, dtype=BF, device=dev)
    w_dg = torch.empty(C, 9 * N, dtype=BF, device=dev)
    ops.conv_weight_layouts(wd, w_fwd, w_dg)
    out_b = torch.empty(B, H, W, N, dtype=BF, device=dev)
    out_f = torch.empty(B, H, W, N, dtype=torch.float32, device=dev)
    ops.conv3x3(xd, w_fwd, out_b, out_f)
    close(out_f, y, rtol=2e-3)
    close(out_b, y)
    if N % 64 == 0:
        dx = torch.empty(B, H, W, C, dtype=BF, device=dev)
        ops.conv3x3(dyd, w_dg, dx, None, flip=True)
        close(dx, x.grad)
        dw = torch.full((3, 3, C, N), 7.0, dtype=torch.float32, device=dev)
        ops.wgrad(xd, dyd, dw)
        close(dw, w.grad, rtol=2e-3)


def test_wgrad_many_pixels_splitk(ops):
    B, H, W, C, N = 4, 32, 32, 64, 64
    x = rnd(B, H, W, C, seed=4).requires_grad_()
    w = rnd(3, 3, C, N, seed=5, scale=0.05).requires_grad_()
    dy = rnd(B, H, W, N, seed=6)
    conv_ref(x, w).backward(dy)
    dw = torch.empty(3, 3, C, N, dtype=torch.float32, device="cuda")
    ops.wgrad(x.detach().cuda().to(BF), dy.cuda().to(BF), dw)
    close(dw, w.grad, rtol=2e-3)


@pytest.mark.parametrize("B,H,W,C,N", [(2, 5, 112, 64, 64), (1, 3, 224, 64, 128), (3, 1, 112, 64, 64),
                                       (5, 40, 224, 64, 64), (2, 4, 112, 128, 128)])
def test_wgrad_row_segment_kernel(ops, B, H, W, C, N):
    """Up to 128 input channels on images whose width is a multiple of 112: the filter-row
    kernel (segments of one image row with a one-pixel halo, three taps per workgroup),
    including one-row images (the rows above and below do not exist) and many slices."""
    x = rnd(B, H, W, C, seed=90).requires_grad_()
    w = rnd(3, 3, C, N, seed=91, scale=0.05).requires_grad_()
    dy = rnd(B, H, W, N, seed=92)
    conv_ref(x, w).backward(dy)
    dw = torch.full((3, 3, C, N), 7.0, dtype=torch.float32, device="cuda")
    ops.wgrad(x.detach().cuda().to(BF), dy.cuda().to(BF), dw)
    close(dw, w.grad, rtol=2e-3)


@pytest.mark.parametrize("B,H,W,C,N", [(2, 56, 56, 128, 64), (3, 28, 28, 64, 128), (3, 14, 14, 256, 64),
                                       (1, 3, 28, 64, 64), (40, 14, 14, 128, 128), (5, 1, 30, 64, 64)])
def test_wgrad_row_group_kernel(ops, B, H, W, C, N):
    """Narrow images: the filter-row kernel whose K-step is a group of whole image rows laid out
    with zero halo entries; 64- and 128-channel tiles, a last group that is not full, images of a
    single row (no rows above or below), groups spanning several images, many slices."""
    x = rnd(B, H, W, C, seed=93).requires_grad_()
    w = rnd(3, 3, C, N, seed=94, scale=0.05).requires_grad_()
    dy = rnd(B, H, W, N, seed=95)
    conv_ref(x, w).backward(dy)
    dw = torch.full((3, 3, C, N), 7.0, dtype=torch.float32, device="cuda")
    ops.wgrad(x.detach().cuda().to(BF), dy.cuda().to(BF), dw)
    close(dw, w.grad, rtol=2e-3)


@pytest.mark.parametrize("M,N,K", [(5, 6, 128), (130, 256, 192), (128, 64, 64), (1, 4096, 256)])
def test_gemm_nt_and_fc_wgrad(ops, M, N, K):
    a, b = rnd(M, K, seed=7), rnd(N, K, seed=8, scale=0.1)
    want = a @ b.t()
    out_f = torch.empty(M, N, dtype=torch.float32, device="cuda")
    out_b = torch.empty(M, N, dtype=BF, device="cuda")
    ops.gemm_nt(a.cuda().to(BF), b.cuda().to(BF), out_b, out_f)
    close(out_f, want, rtol=2e-3)
    close(out_b, want)
    if N % 64 == 0:
        dy = rnd(M, N, seed=9)
        dw = torch.empty(K, N, dtype=torch.float32, device="cuda")
        ops.wgrad(a.cuda().to(BF), dy.cuda().to(BF), dw)
        close(dw, a.t() @ dy, rtol=2e-3)


def test_first_layer(ops):
    B, H, W, C, N = 2, 10, 12, 4, 64
    x = rnd(B, H, W, C, seed=10).requires_grad_()
    w = rnd(3, 3, C, N, seed=11, scale=0.05).requires_grad_()   # bfloat16-representable: the kernel rounds the filter
    y = conv_ref(x, w)
    dy = rnd(B, H, W, N, seed=12)
    y.backward(dy)
    out = torch.empty(B, H, W, N, dtype=BF, device="cuda")
    xd = x.detach().cuda().to(BF)
    ops.conv3x3_small(xd, w.detach().cuda(), out)
    close(out, y)
    dw = torch.empty(3, 3, C, N, dtype=torch.float32, device="cuda")
    ops.conv3x3_small_wgrad(xd, dy.cuda().to(BF), dw)
    close(dw, w.grad, rtol=2e-3)


@pytest.mark.parametrize("P_shape,C", [((2, 6, 6), 64), ((3, 5, 7), 128), ((1, 3, 3), 512)])
def test_batchnorm_relu(ops, P_shape, C):
    eps = 1e-3
    z = rnd(*P_shape, C, seed=13).requires_grad_()
    gamma = (1.0 + 0.1 * rnd(C, seed=14)).requires_grad_()
    beta = (0.1 * rnd(C, seed=15)).requires_grad_()
    mean = z.mean(dim=(0, 1, 2))
    var = z.var(dim=(0, 1, 2), unbiased=False)
    y = torch.relu((z - mean) * torch.rsqrt(var + eps) * gamma + beta)
    dy = rnd(*P_shape, C, seed=16)
    y.backward(dy)
    dev = "cuda"
    zd = z.detach().to(dev).to(BF)
    sums = torch.empty(2, C, dtype=torch.float32, device=dev)
    ops.channel_stats(zd, sums)
    n = z.numel() // C
    close(sums[0] / n, mean, rtol=1e-4, atol=1e-5)
    close(sums[1] / n - (sums[0] / n) ** 2, var, rtol=1e-3)
    yd = torch.empty_like(zd)
    ops.bn_relu(zd, sums, gamma.detach().to(dev), beta.detach().to(dev), yd, eps)
    close(yd, y)
    # the backward kernel gates on the stored bfloat16 y: feed the reference's own mask
    # through the same rounding by using yd as produced above
    red = torch.empty(2, C, dtype=torch.float32, device=dev)
    dz = torch.empty_like(zd)
    ops.bn_relu_bwd(dy.to(dev).to(BF), zd, sums, gamma.detach().to(dev), beta.detach().to(dev), red, dz, eps)
    close(red[0], beta.grad, rtol=5e-3)
    close(red[1], gamma.grad, rtol=5e-3)
    close(dz, z.grad, rtol=2e-2)


def test_maxpool(ops):
    B, H, W, C = 2, 6, 8, 64
    x = rnd(B, H, W, C, seed=17).requires_grad_()
    y = F.max_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
    dy = rnd(B, H // 2, W // 2, C, seed=18)
    y.backward(dy)
    xd = x.detach().cuda().to(BF)
    yd = torch.empty(B, H // 2, W // 2, C, dtype=BF, device="cuda")
    ops.maxpool2(xd, yd)
    assert torch.equal(yd.float().cpu(), y.detach())
    dx = torch.empty_like(xd)
    ops.maxpool2_bwd(xd, yd, dy.cuda().to(BF), dx)
    assert torch.equal(dx.float().cpu(), x.grad)


def test_maxpool_ties_go_to_first(ops):
    x = torch.ones(1, 2, 2, 8, dtype=BF, device="cuda")
    y = torch.empty(1, 1, 1, 8, dtype=BF, device="cuda")
    ops.maxpool2(x, y)
    dx = torch.empty_like(x)
    ops.maxpool2_bwd(x, y, torch.full_like(y, 3.0), dx)
    assert torch.equal(dx.float().cpu().reshape(4, 8)[:, 0], torch.tensor([3.0, 0.0, 0.0, 0.0]))


def test_bias_act_loss_adam_casts(ops):
    M, N = 7, 128
    z, b = rnd(M, N, seed=19), rnd(N, seed=20)
    yb = torch.empty(M, N, dtype=BF, device="cuda")
    yf = torch.empty(M, N, dtype=torch.float32, device="cuda")
    ops.bias_act(z.cuda(), b.cuda(), yb, yf, relu=True)
    want = torch.relu(z + b)
    close(yf, want, rtol=1e-6)
    close(yb, want)
    dy = rnd(M, N, seed=21)
    dz = torch.empty(M, N, dtype=BF, device="cuda")
    db = torch.empty(N, dtype=torch.float32, device="cuda")
    ops.bias_act_bwd(dy.cuda(), yb, dz, db, relu=True)
    g = dy * (want > 0)
    close(dz, g)
    close(db, g.sum(0), rtol=1e-5)
    ops.bias_act_bwd(dy.cuda(), None, dz, db, relu=False)
    close(db, dy.sum(0), rtol=1e-5)

    # loss: 0.5 * sum(((out - gt) Wl)^2) and its gradient
    Bn, D = 9, 6
    out = rnd(Bn, D, seed=22).requires_grad_()
    gt, Wl = rnd(Bn, D, seed=23), rnd(D, D, seed=24)
    loss = 0.5 * (((out - gt) @ Wl) ** 2).sum()
    loss.backward()
    ld = torch.empty(1, dtype=torch.float32, device="cuda")
    dod = torch.empty(Bn, D, dtype=torch.float32, device="cuda")
    ops.l2loss(out.detach().cuda(), gt.cuda(), Wl.cuda(), ld, dod)
    close(ld, loss.reshape(1), rtol=1e-5)
    close(dod, out.grad, rtol=1e-4)

    # Adam against torch.optim.Adam (same update rule as TensorFlow's up to where eps enters;
    # checked against the TensorFlow formula written out)
    n = 1000
    p0, g0 = rnd(n, seed=25), rnd(n, seed=26)
    p, m, v = p0.clone().cuda(), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    pr, mr, vr = p0.clone().double(), torch.zeros(n).double(), torch.zeros(n).double()
    for step in (1, 2, 3):
        g = g0 * step
        ops.adam_step(p, g.cuda(), m, v, lr=0.01, step=step)
        mr = 0.9 * mr + 0.1 * g.double()
        vr = 0.999 * vr + 0.001 * g.double() ** 2
        lr_t = 0.01 * np.sqrt(1 - 0.999 ** step) / (1 - 0.9 ** step)
        pr = pr - lr_t * mr / (vr.sqrt() + 1e-8)
    close(p, pr.float(), rtol=1e-5)

    src = rnd(37, 70, seed=27)
    dst = torch.empty(70, 37, dtype=BF, device="cuda")
    ops.transpose_cast(src.cuda(), dst)
    assert torch.equal(dst.float().cpu(), src.t())
    dst2 = torch.empty(37, 70, dtype=BF, device="cuda")
    ops.cast_bf16(src.cuda(), dst2)
    assert torch.equal(dst2.float().cpu(), src)


def test_alternative_losses_against_the_oracle(ops):
    from oracle import slam_oracle as so
    from slam_amd.network import losses
    B = 5
    out, gt, Wl = rnd(B, 6, seed=30), rnd(B, 6, seed=31), rnd(6, 6, seed=32)
    loss, d = losses.lstm_loss(out.cuda(), gt.cuda(), Wl.cuda())
    assert abs(float(loss) - so.lstm_loss(out.numpy(), gt.numpy(), Wl.numpy())) <= 1e-5 * abs(float(loss))
    o = out.clone().requires_grad_()
    ((((o - gt) @ Wl) ** 2).sum() / B).backward()
    close(d, o.grad, rtol=1e-4)
    out7, gt7 = rnd(B, 7, seed=33), rnd(B, 7, seed=34)
    loss, d = losses.posenet_loss(out7.cuda(), gt7.cuda())
    want = so.posenet_loss(out7.numpy(), gt7.numpy())
    assert abs(float(loss) - want) <= 1e-5 * abs(want)
    o = out7.clone().requires_grad_()
    (((o[:, :3] - gt7[:, :3]) ** 2).sum() + 1100 * ((o[:, 3:] - gt7[:, 3:]) ** 2).sum()).backward()
    close(d, o.grad, rtol=1e-5)
    total, grads = losses.posenet_multi_head_loss([out7.cuda()] * 3, gt7.cuda())
    assert abs(float(total) - 1.6 * want) <= 1e-4 * abs(want)


@pytest.mark.parametrize("shape,C", [((2, 5, 6), 64), ((1, 3, 3), 256)])
def test_bias_relu_layer(ops, shape, C):
    z = rnd(*shape, C, seed=35).requires_grad_()
    b = rnd(C, seed=36).requires_grad_()
    y = torch.relu(z + b)
    dy = rnd(*shape, C, seed=37)
    y.backward(dy)
    zd = z.detach().cuda().to(BF)
    yd = torch.empty_like(zd)
    ops.bias_relu(zd, b.detach().cuda(), yd)
    close(yd, y)
    dz = torch.empty_like(zd)
    db = torch.full((C,), 3.0, device="cuda")
    ops.bias_relu_bwd(dy.cuda().to(BF), yd, dz, db)
    close(dz, z.grad)
    close(db, b.grad, rtol=5e-3)


@pytest.mark.parametrize("B,H,W,C,N", [(2, 9, 11, 64, 64), (1, 14, 14, 128, 256), (3, 10, 12, 4, 64)])
def test_conv_with_fused_statistics(ops, B, H, W, C, N):
    """The statistics from the convolution's epilogue equal those of a separate pass over
    the stored bfloat16 output."""
    x = rnd(B, H, W, C, seed=40).cuda().to(BF)
    w = rnd(3, 3, C, N, seed=41, scale=0.05).cuda()
    out = torch.empty(B, H, W, N, dtype=BF, device="cuda")
    sums = torch.full((2, N), 9.0, device="cuda")
    ws = torch.empty(max(ops.conv_stats_workspace_bytes(B * H * W, N), 16) // 4, device="cuda")
    if C == 4:
        ops.conv3x3_small_stats(x, w, out, sums, ws)
        ref_out = torch.empty_like(out)
        ops.conv3x3_small(x, w, ref_out)
    else:
        w_fwd = torch.empty(N, 9 * C, dtype=BF, device="cuda")
        ops.conv_weight_layouts(w, w_fwd, None)
        ops.conv3x3_stats(x, w_fwd, out, sums, ws)
        ref_out = torch.empty_like(out)
        ops.conv3x3(x, w_fwd, out_bf16=ref_out)
    assert torch.equal(out, ref_out)
    z = out.float().reshape(-1, N).double()
    close(sums[0], z.sum(0).float().cpu(), rtol=1e-5, atol=1e-3)
    close(sums[1], (z * z).sum(0).float().cpu(), rtol=1e-5, atol=1e-3)


@pytest.mark.parametrize("M,N,K", [(4, 256, 4096), (128, 64, 2048), (3, 6, 1024)])
def test_gemm_split_k(ops, M, N, K):
    """Few output tiles and a long K: the float32-only call splits K over workgroups."""
    a, b = rnd(M, K, seed=50), rnd(N, K, seed=51, scale=0.1)
    out = torch.full((M, N), 5.0, dtype=torch.float32, device="cuda")
    ops.gemm_nt(a.cuda().to(BF), b.cuda().to(BF), out_f32=out)
    close(out, a @ b.t(), rtol=2e-3)


@pytest.mark.parametrize("B", [16, 17])
def test_wide_layer_uses_256_tiles(ops, B):
    """A layer wide and large enough for the 256 x 256-tile kernel (N % 256 == 0, >= 256 tiles):
    forward with fused statistics and data gradient against float32 on the CPU.  B = 16 gives
    exactly one tile per compute unit; B = 17 leaves 16 tiles over, which go to the 128-tile
    kernel (pixel-tile offset, statistics records behind the first kernel's)."""
    H, W, C, N = 64, 64, 256, 256
    x = rnd(B, H, W, C, seed=60, scale=0.5).requires_grad_()
    w = rnd(3, 3, C, N, seed=61, scale=0.02).requires_grad_()
    y = conv_ref(x, w)
    dy = rnd(B, H, W, N, seed=62)
    y.backward(dy)
    xd, dyd = x.detach().cuda().to(BF), dy.cuda().to(BF)
    w_fwd, w_dg = torch.empty(N, 9 * C, dtype=BF, device="cuda"), torch.empty(C, 9 * N, dtype=BF, device="cuda")
    ops.conv_weight_layouts(w.detach().cuda(), w_fwd, w_dg)
    out = torch.empty(B, H, W, N, dtype=BF, device="cuda")
    sums = torch.empty(2, N, device="cuda")
    ws = torch.empty(ops.conv_stats_workspace_bytes(B * H * W, N) // 4, device="cuda")
    ops.conv3x3_stats(xd, w_fwd, out, sums, ws)
    close(out, y)
    z = out.float().reshape(-1, N).double()
    close(sums[0], z.sum(0).float().cpu(), rtol=1e-5, atol=1e-2)
    close(sums[1], (z * z).sum(0).float().cpu(), rtol=1e-5, atol=1e-2)
    out2 = torch.empty_like(out)
    ops.conv3x3(xd, w_fwd, out_bf16=out2)
    assert torch.equal(out, out2)
    dx = torch.empty(B, H, W, C, dtype=BF, device="cuda")
    ops.conv3x3(dyd, w_dg, out_bf16=dx, flip=True)
    close(dx, x.grad)


@pytest.mark.parametrize("B,H,W,C", [(2, 8, 12, 64), (3, 14, 14, 512), (1, 6, 4, 192)])
def test_fused_bn_relu_pool_matches_the_separate_kernels(B, H, W, C):
    """Normalisation + ReLU + 2x2 max-pool in one pass, and pool backward + gate + batch-norm
    backward in one step, against the separate kernels (identical forward bits; the backward
    differs only by the summation order of the two channel reductions)."""
    from slam_amd.network import encoder_ops as ops
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(B * H + C)
    z = torch.randn(B, H, W, C, generator=g).to(dev).to(torch.bfloat16)
    z[0, :2, :2, :8] = z[0, 0, 0, :8]                      # a window of equal values: the first one wins
    gamma = (1 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    sums = torch.empty(2, C, device=dev)
    ops.channel_stats(z, sums)
    y_ref, y = torch.empty_like(z), torch.empty_like(z)
    pooled_ref = torch.empty(B, H // 2, W // 2, C, dtype=torch.bfloat16, device=dev)
    pooled = torch.empty_like(pooled_ref)
    ops.bn_relu(z, sums, gamma, beta, y_ref, 1e-3)
    ops.maxpool2(y_ref, pooled_ref)
    ops.bn_relu_pool(z, sums, gamma, beta, y, pooled, 1e-3)
    assert torch.equal(y, y_ref) and torch.equal(pooled, pooled_ref)
    pooled_only = torch.empty_like(pooled_ref)
    ops.bn_relu_pool(z, sums, gamma, beta, None, pooled_only, 1e-3)     # y is optional
    assert torch.equal(pooled_only, pooled_ref)
    dp = torch.randn(B, H // 2, W // 2, C, generator=g).to(dev).to(torch.bfloat16)
    dy = torch.empty_like(z)
    ops.maxpool2_bwd(y_ref, pooled_ref, dp, dy)
    red_ref, red = torch.empty(2, C, device=dev), torch.empty(2, C, device=dev)
    dz_ref, dz = torch.empty_like(z), torch.empty_like(z)
    ops.bn_relu_bwd(dy, z, sums, gamma, beta, red_ref, dz_ref, 1e-3)
    ops.bn_relu_pool_bwd(dp, z, sums, gamma, beta, red, dz, 1e-3)
    np.testing.assert_allclose(red.cpu().numpy(), red_ref.cpu().numpy(), rtol=1e-4, atol=1e-4)
    a, b = dz.float().cpu().numpy(), dz_ref.float().cpu().numpy()
    assert np.abs(a - b).max() <= 0.02 * np.abs(b).max() and (a != b).mean() < 0.02


@pytest.mark.parametrize("B,H,W,C,N", [(3, 224, 224, 64, 64), (2, 260, 256, 128, 64), (11, 112, 112, 128, 64),
                                       (3, 211, 208, 64, 64),   # pixels not a multiple of the 256-row tile
                                       (11, 112, 112, 64, 128), (11, 112, 112, 128, 128),   # two 64-wide tiles
                                       (42, 56, 56, 256, 128)])
def test_filter_row_convolution_kernel(ops, B, H, W, C, N):
    """Layers with many pixels (>= 128 Ki), at most 128 channels per tap and 64 outputs take the filter-row
    kernel (one K-step per filter row, taps read the staged tile at row offsets, validity as
    per-lane fragment masks): forward with fused statistics and data gradient against float32."""
    x = rnd(B, H, W, C, seed=80, scale=0.5).requires_grad_()
    w = rnd(3, 3, C, N, seed=81, scale=0.05).requires_grad_()
    y = conv_ref(x, w)
    dy = rnd(B, H, W, N, seed=82)
    y.backward(dy)
    xd, dyd = x.detach().cuda().to(BF), dy.cuda().to(BF)
    w_fwd, w_dg = torch.empty(N, 9 * C, dtype=BF, device="cuda"), torch.empty(C, 9 * N, dtype=BF, device="cuda")
    ops.conv_weight_layouts(w.detach().cuda(), w_fwd, w_dg)
    out = torch.empty(B, H, W, N, dtype=BF, device="cuda")
    sums = torch.empty(2, N, device="cuda")
    ws = torch.empty(ops.conv_stats_workspace_bytes(B * H * W, N) // 4, device="cuda")
    ops.conv3x3_stats(xd, w_fwd, out, sums, ws)
    close(out, y)
    z = out.float().reshape(-1, N).double()
    close(sums[0], z.sum(0).float().cpu(), rtol=1e-5, atol=1e-2)
    close(sums[1], (z * z).sum(0).float().cpu(), rtol=1e-5, atol=1e-2)
    out2 = torch.empty_like(out)
    ops.conv3x3(xd, w_fwd, out_bf16=out2)
    assert torch.equal(out, out2)
    dx = torch.empty(B, H, W, C, dtype=BF, device="cuda")
    ops.conv3x3(dyd, w_dg, out_bf16=dx, flip=True)
    close(dx, x.grad)


@pytest.mark.gpu
@pytest.mark.parametrize("R,C", [(128, 64), (25088 // 8, 4096 // 8), (200, 72), (37, 19)])
def test_transpose_bf16(ops, R, C):
    """bf16 [R][C] -> [C][R]: the 16-byte path (R, C multiples of 8) and the ragged fallback."""
    g = torch.Generator().manual_seed(R * 1000 + C)
    src = torch.randn(R, C, generator=g).to(torch.bfloat16).cuda()
    dst = torch.zeros(C, R, dtype=torch.bfloat16, device="cuda")
    ops.transpose_bf16(src, dst)
    torch.cuda.synchronize()
    assert torch.equal(dst.cpu(), src.cpu().t().contiguous())


@pytest.mark.gpu
def test_adam_shadow_matches_plain_update(ops):
    """The shadow-writing Adam step updates exactly as the plain one and leaves bf16(p) of the
    leading parameters in the shadow; the rest of the shadow buffer is not touched."""
    n, ns = 10_000, 6_400
    g = torch.Generator().manual_seed(5)
    p0 = torch.randn(n, generator=g)
    grad = torch.randn(n, generator=g)
    pa, pb = p0.clone().cuda(), p0.clone().cuda()
    ma, va = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    mb, vb = torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    shadow = torch.full((ns + 64,), 7.0, dtype=torch.bfloat16, device="cuda")
    for step in (1, 2, 3):
        ops.adam_step(pa, grad.cuda(), ma, va, lr=0.01, step=step, grad_scale=0.5)
        ops.adam_step(pb, grad.cuda(), mb, vb, lr=0.01, step=step, grad_scale=0.5, shadow=shadow[:ns])
    torch.cuda.synchronize()
    assert torch.equal(pa, pb) and torch.equal(ma, mb) and torch.equal(va, vb)
    assert torch.equal(shadow[:ns].cpu(), pb[:ns].to(torch.bfloat16).cpu())
    assert bool((shadow[ns:] == 7.0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 10, 12), (3, 32, 32), (1, 7, 9)])
def test_first_layer_backward_fused(ops, B, H, W):
    """Batch-norm backward + filter gradient of the first layer without the dz tensor against
    the two separate calls (same arithmetic; the filter gradient differs only by the order of
    its float atomics)."""
    g = torch.Generator().manual_seed(B * 100 + H)
    dev = "cuda"
    x = torch.randn(B, H, W, 4, generator=g).to(dev).to(torch.bfloat16)
    z = torch.randn(B, H, W, 64, generator=g).to(dev).to(torch.bfloat16)
    dy = torch.randn(B, H, W, 64, generator=g).to(dev).to(torch.bfloat16)
    gamma = (1.0 + 0.1 * torch.randn(64, generator=g)).to(dev)
    beta = (0.1 * torch.randn(64, generator=g)).to(dev)
    sums = torch.empty(2, 64, device=dev)
    ops.channel_stats(z, sums)
    red_ref, red = torch.empty(2, 64, device=dev), torch.full((2, 64), 7.0, device=dev)
    dz = torch.empty_like(z)
    dw_ref, dw = torch.empty(3, 3, 4, 64, device=dev), torch.full((3, 3, 4, 64), 7.0, device=dev)
    ops.bn_relu_bwd(dy, z, sums, gamma, beta, red_ref, dz, 1e-3)
    ops.conv3x3_small_wgrad(x, dz, dw_ref)
    ops.bn_relu_bwd_first_layer(dy, z, sums, gamma, beta, red, x, dw, 1e-3)
    torch.cuda.synchronize()
    np.testing.assert_allclose(red.cpu().numpy(), red_ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(dw.cpu().numpy(), dw_ref.cpu().numpy(), rtol=1e-4, atol=1e-4 * float(dw_ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("B,H,W", [(2, 10, 12), (3, 32, 32), (1, 7, 9), (4, 64, 64)])
def test_first_layer_without_stored_z(ops, B, H, W):
    """The first layer's forward that writes y alone, and its backward that recomputes z from the input, against
    the calls that go through a stored z (cnn_model.py:100-113): y bit for bit (same statistics), sums, red and the
    filter gradient up to the order of their float sums."""
    g = torch.Generator().manual_seed(B * 1000 + H)
    dev = "cuda"
    x = torch.randn(B, H, W, 4, generator=g).to(dev).to(torch.bfloat16)
    w = (0.5 * torch.randn(3, 3, 4, 64, generator=g)).to(dev)
    dy = torch.randn(B, H, W, 64, generator=g).to(dev).to(torch.bfloat16)
    gamma = (1.0 + 0.1 * torch.randn(64, generator=g)).to(dev)
    beta = (0.1 * torch.randn(64, generator=g)).to(dev)
    ws = torch.empty(max(ops.conv_stats_workspace_bytes(B * H * W, 64), 1 << 20) // 4, device=dev)
    # through z
    z = torch.empty(B, H, W, 64, dtype=torch.bfloat16, device=dev)
    sums_ref = torch.empty(2, 64, device=dev)
    ops.conv3x3_small_stats(x, w, z, sums_ref, ws)
    # without z
    y = torch.full((B, H, W, 64), 7.0, dtype=torch.bfloat16, device=dev)
    sums = torch.full((2, 64), 7.0, device=dev)
    ops.conv3x3_c4_bn_relu(x, w, gamma, beta, y, sums, ws, 1e-3)
    y_ref = torch.empty_like(y)
    ops.bn_relu(z, sums, gamma, beta, y_ref, 1e-3)          # the same statistics: the same y
    torch.cuda.synchronize()
    np.testing.assert_allclose(sums.cpu().numpy(), sums_ref.cpu().numpy(), rtol=1e-5, atol=1e-4)
    assert torch.equal(y.cpu(), y_ref.cpu())
    red_ref, red = torch.empty(2, 64, device=dev), torch.full((2, 64), 7.0, device=dev)
    dw_ref, dw = torch.empty(3, 3, 4, 64, device=dev), torch.full((3, 3, 4, 64), 7.0, device=dev)
    ops.bn_relu_bwd_first_layer(dy, z, sums, gamma, beta, red_ref, x, dw_ref, 1e-3)
    ops.bn_relu_bwd_first_layer_recompute(dy, x, w, sums, gamma, beta, red, dw, ws, 1e-3)
    torch.cuda.synchronize()
    scale = float(red_ref.abs().max())
    np.testing.assert_allclose(red.cpu().numpy(), red_ref.cpu().numpy(), rtol=1e-4, atol=1e-5 * scale)
    np.testing.assert_allclose(dw.cpu().numpy(), dw_ref.cpu().numpy(), rtol=1e-3, atol=1e-3 * float(dw_ref.abs().max()))


@pytest.mark.gpu
@pytest.mark.parametrize("pool", [False, True])
def test_batch_norm_backward_in_two_phases_over_two_shards_matches_the_whole_batch(ops, pool):
    """What two ranks with shared statistics compute for one layer (cnn_model.py:100-113 on the whole batch): each
    shard runs the reduction pass on its half, the sums are averaged (every rank divides by its OWN row count, so
    the mean of the ranks' sums stands for the whole batch's), each shard runs the apply pass - against the one-call
    backward on the whole batch.  Tight, because a single layer does not amplify rounding."""
    g = torch.Generator().manual_seed(77)
    dev = "cuda"
    B, H, W, C = 4, 16, 24, 128
    z = torch.randn(B, H, W, C, generator=g).to(dev).to(torch.bfloat16)
    gamma = (1.0 + 0.1 * torch.randn(C, generator=g)).to(dev)
    beta = (0.1 * torch.randn(C, generator=g)).to(dev)
    dy_shape = (B, H // 2, W // 2, C) if pool else (B, H, W, C)
    dy = torch.randn(*dy_shape, generator=g).to(dev).to(torch.bfloat16)
    bwd = ops.bn_relu_pool_bwd if pool else ops.bn_relu_bwd
    sums = torch.empty(2, C, device=dev)
    ops.channel_stats(z, sums)
    red_ref, dz_ref = torch.empty(2, C, device=dev), torch.empty_like(z)
    bwd(dy, z, sums, gamma, beta, red_ref, dz_ref, 1e-3)
    # two shards: statistics and backward sums averaged, as dp.average_over_ranks does
    h = B // 2
    shard_sums = [torch.empty(2, C, device=dev) for _ in range(2)]
    for r in range(2):
        ops.channel_stats(z[r * h:(r + 1) * h].contiguous(), shard_sums[r])
    mean_sums = (shard_sums[0] + shard_sums[1]) / 2
    reds = [torch.empty(2, C, device=dev) for _ in range(2)]
    dz = torch.empty_like(z)
    parts = []
    for r in range(2):
        zs, dys = z[r * h:(r + 1) * h].contiguous(), dy[r * h:(r + 1) * h].contiguous()
        out = torch.empty_like(zs)
        bwd(dys, zs, mean_sums, gamma, beta, reds[r], out, 1e-3, phases=1)
        parts.append((dys, zs, out))
    mean_red = (reds[0] + reds[1]) / 2
    for r, (dys, zs, out) in enumerate(parts):
        bwd(dys, zs, mean_sums, gamma, beta, mean_red.clone(), out, 1e-3, phases=2)
        dz[r * h:(r + 1) * h] = out
    torch.cuda.synchronize()
    np.testing.assert_allclose(2 * mean_sums.cpu().numpy(), sums.cpu().numpy(), rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(2 * mean_red.cpu().numpy(), red_ref.cpu().numpy(), rtol=1e-3, atol=2e-2)
    err = float((dz.float() - dz_ref.float()).norm() / dz_ref.float().norm())
    assert err < 1e-2, err
