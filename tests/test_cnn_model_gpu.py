"""The whole encoder (forward, loss, backward, optimizer step) on the GPU against
the same network written in plain PyTorch float32 on the CPU with autograd.
Unpinned by the reference (TensorFlow 0.x and its weights file are absent); the
network definition follows slam/network/cnn_model.py:31-80,139-166.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def torch_reference(model, x, gt, Wl, eps, storage_rounding):
    """float32 autograd restatement (oracle/torch_encoder.py); parameters are cloned from the model."""
    from oracle import torch_encoder
    P = {k: v.detach().cpu().clone().requires_grad_() for k, v in model.params.items()}
    out, loss = torch_encoder.forward_loss(P, x, gt, Wl, eps, storage_rounding)
    loss.backward()
    return out.detach(), loss.detach(), P


def rel_l2(got, want):
    got, want = got.detach().float().cpu().flatten(), want.detach().float().flatten()
    return float((got - want).norm() / (want.norm() + 1e-20))


@pytest.fixture(scope="module")
def small_model():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slam_amd.network.cnn_model import VGG16Model
    B, H, W = 4, 64, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, 4, generator=g).to(torch.bfloat16).float()
    inp = x.cuda()
    model = VGG16Model(B, inp, 6, 1e-3, fc_dim=256, seed=1)
    model.build_graph()
    gt = torch.randn(B, 6, generator=g)
    Wl = torch.eye(6)
    model.add_loss(Wl.cuda(), gt.cuda())
    return model, x, gt, Wl


def test_forward_layer_by_layer(small_model):
    """Every layer, fed the GPU's own input of that layer, against float32 on the CPU with
    the same bfloat16 storage points.  (End to end the randomly initialised network is
    chaotic at bfloat16 resolution - one-ulp flips in 0.1 % of the activations move the
    output by ~4 % in the CPU restatement itself - so the tight check is per layer.)"""
    from slam_amd.network.cnn_model import CONV_LAYERS
    model, x, gt, Wl = small_model
    model.keep_full_y = True     # pooled layers write their full-resolution output only on request
    model.first_layer_recompute = False     # ... and the first layer its z
    model.forward()
    model.keep_full_y = False
    model.first_layer_recompute = True
    torch.cuda.synchronize()
    P = {k: v.detach().cpu() for k, v in model.params.items()}

    def q(t):
        return t.to(torch.bfloat16).float()

    h = x.permute(0, 3, 1, 2)
    for i, (name, _, pool) in enumerate(CONV_LAYERS):
        a = model._act[i]
        w = P[name + "/weights"]
        z = q(F.conv2d(h, q(w).permute(3, 2, 0, 1), padding=1))
        assert rel_l2(a["z"].permute(0, 3, 1, 2), z) < 1e-3, name
        zg = a["z"].float().cpu().permute(0, 3, 1, 2)
        mean = zg.mean(dim=(0, 2, 3), keepdim=True)
        var = zg.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        bn = P[name + "/bn"]
        y = q(torch.relu((zg - mean) * torch.rsqrt(var + 1e-3) * bn[1].view(1, -1, 1, 1) + bn[0].view(1, -1, 1, 1)))
        assert rel_l2(a["y"].permute(0, 3, 1, 2), y) < 1e-3, name
        if pool:
            yg = a["y"].float().cpu().permute(0, 3, 1, 2)
            assert torch.equal(a["pooled"].float().cpu().permute(0, 3, 1, 2), F.max_pool2d(yg, 2)), name
        h = (a["pooled"] if pool else a["y"]).float().cpu().permute(0, 3, 1, 2)
    feat = h.permute(0, 2, 3, 1).reshape(x.shape[0], -1)       # NHWC flatten (tf.reshape)
    a6 = q(torch.relu(feat @ q(P["fc6-conv/weights"]) + P["fc6-conv/bias"]))
    assert rel_l2(model._fc["fc6-conv"]["a"], a6) < 1e-3
    a6g = model._fc["fc6-conv"]["a"].float().cpu()
    a7 = q(torch.relu(a6g @ q(P["fc7-conv/weights"]) + P["fc7-conv/bias"]))
    assert rel_l2(model._fc["fc7-conv"]["a"], a7) < 1e-3
    a7g = model._fc["fc7-conv"]["a"].float().cpu()
    out = a7g @ q(P["out_layer/weights"])[:, :6] + P["out_layer/bias"][:6]
    assert rel_l2(model.network_output, out) < 1e-4


def test_backward_layer_by_layer(small_model):
    """Each layer's backward against autograd of the same layer on the CPU, fed the GPU's
    own stored input of the layer and the GPU's own incoming gradient."""
    from slam_amd.network.cnn_model import CONV_LAYERS
    model, x, gt, Wl = small_model
    model.trace = {}
    model.forward()
    model.compute_loss()
    model.backward()
    torch.cuda.synchronize()
    trace, model.trace = model.trace, None
    P = {k: v.detach().cpu() for k, v in model.params.items()}
    B = x.shape[0]

    def q(t):
        return t + (t.to(torch.bfloat16).float() - t).detach()

    def cpu(t):
        return t.detach().float().cpu()

    # loss gradient and the fully connected layers
    out = cpu(model.network_output)
    assert rel_l2(model._d_out, (out - gt) @ Wl @ Wl.t()) < 1e-5
    d = cpu(model._d_out_pad)
    feat = cpu(model._act[-1]["pooled"]).reshape(B, -1)
    ins = {"out_layer": cpu(model._fc["fc7-conv"]["a"]), "fc7-conv": cpu(model._fc["fc6-conv"]["a"]), "fc6-conv": feat}
    for n in ("out_layer", "fc7-conv", "fc6-conv"):
        a_in = ins[n].clone().requires_grad_()
        w = P[n + "/weights"].clone().requires_grad_()
        b = P[n + "/bias"].clone().requires_grad_()
        zz = a_in @ q(w) + b
        yy = zz if n == "out_layer" else torch.relu(zz)
        yy.backward(d)
        assert rel_l2(model.grads[n + "/weights"], w.grad) < 1e-2, n
        assert rel_l2(model.grads[n + "/bias"], b.grad) < 1e-2, n
        if n == "fc6-conv":
            assert rel_l2(trace["d_feat"].reshape(B, -1), a_in.grad) < 1e-2, n
        else:
            assert rel_l2(model._fc[n]["d_in"], a_in.grad) < 1e-2, n
            d = cpu(model._fc[n]["d_in"])

    for i in range(len(CONV_LAYERS) - 1, -1, -1):
        name, _, pool = CONV_LAYERS[i]
        a = model._act[i]
        if i == 0:
            h_in = x.clone()
        else:
            prev = model._act[i - 1]
            h_in = cpu(prev["pooled"] if prev["pool"] else prev["y"])
        h_in = h_in.permute(0, 3, 1, 2).clone().requires_grad_()
        w = P[name + "/weights"].clone().requires_grad_()
        bn = P[name + "/bn"].clone().requires_grad_()
        wq = q(w).permute(3, 2, 0, 1)
        # (a) normalisation + ReLU (+ pool) backward: dz, dbeta, dscale from the incoming gradient
        z = q(F.conv2d(h_in, wq, padding=1)).detach().requires_grad_()
        mean = z.mean(dim=(0, 2, 3), keepdim=True)
        var = z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        y = q(torch.relu((z - mean) * torch.rsqrt(var + 1e-3) * bn[1].view(1, -1, 1, 1) + bn[0].view(1, -1, 1, 1)))
        top = F.max_pool2d(y, 2) if pool else y     # the pool picks among the stored bfloat16 values
        top.backward(cpu(trace[name]["g_out"]).permute(0, 3, 1, 2))
        assert rel_l2(model.grads[name + "/bn"], bn.grad) < 1e-2, name
        assert rel_l2(trace[name]["dz"].permute(0, 3, 1, 2), z.grad) < 2e-2, name   # stored as bfloat16; rare ReLU-gate flips
        # (b) convolution backward from the dz the GPU stored.  (Against the unrounded dz the
        #     filter gradient is a few percent off: sum(dz) = 0 per channel cancels the large
        #     mean of the post-ReLU input, and bfloat16 rounding of dz breaks that cancellation.)
        F.conv2d(h_in, wq, padding=1).backward(cpu(trace[name]["dz"]).permute(0, 3, 1, 2))
        assert rel_l2(model.grads[name + "/weights"], w.grad) < 2e-3, name
        if i:
            assert rel_l2(trace[name]["d_in"].permute(0, 3, 1, 2), h_in.grad) < 1e-2, name
    # padded output columns never receive gradient
    assert float(model.grads["out_layer/weights"][:, 6:].abs().max()) == 0.0


def test_end_to_end_against_autograd(small_model):
    """Loose: the randomly initialised network is chaotic at bfloat16 resolution, so end to
    end only gross wiring mistakes (errors of order one) are caught here; the tight
    checks are the two layer-by-layer tests above."""
    model, x, gt, Wl = small_model
    out = model.forward()
    model.compute_loss()
    model.backward()
    torch.cuda.synchronize()
    out_ref, loss_ref, P = torch_reference(model, x, gt, Wl, 1e-3, storage_rounding=True)
    e_out = rel_l2(out, out_ref)
    worst = {name: rel_l2(model.grads[name], ref.grad) for name, ref in P.items()}
    print("output rel-L2 %.4f; gradient rel-L2 %s" % (e_out, {k: round(v, 3) for k, v in worst.items()}))
    assert e_out < 0.1
    assert abs(float(model.loss) - float(loss_ref)) <= 0.1 * abs(float(loss_ref))
    bad = {k: v for k, v in worst.items() if not v < 0.6}
    assert not bad, "gradient mismatch: %s" % (bad,)


def test_training_reduces_loss(small_model):
    model, x, gt, Wl = small_model
    step = model.add_optimizer()
    # the reference's schedule: 0.01 * 0.1^(global_step // 5)  (cnn_model.py:149-150)
    lr = model._optimizer.learning_rate
    assert [lr(0), lr(4)] == [0.01, 0.01] and abs(lr(5) - 1e-3) < 1e-12 and abs(lr(10) - 1e-4) < 1e-12
    model._optimizer.learning_rate = 1e-4       # a step size at which 8 steps must descend
    losses = [float(step()) for _ in range(8)]
    assert np.all(np.isfinite(losses)) and losses[-1] < 0.7 * losses[0], losses
    assert model.global_step == 8


def test_zero_loss_weight_gives_zero_gradients(small_model):
    # the training script feeds an all-zero loss weight for the first batch of each
    # sequence (slam/main/vgg16_rgbd.py:51)
    model, x, gt, Wl = small_model
    model._loss_weight.zero_()
    model.forward()
    model.compute_loss()
    model.backward()
    torch.cuda.synchronize()
    assert float(model.loss) == 0.0
    assert float(model.flat_grads.abs().max()) == 0.0
    model._loss_weight.copy_(torch.eye(6))


def test_no_batch_norm_variant_trains():
    """cnn_model_noBN.VGG16Model: conv -> bias -> ReLU; gradients against autograd on the
    output layer's input chain are covered per operator, here the wiring: parameter table,
    finite loss, descent."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slam_amd.network.cnn_model_noBN import VGG16Model
    B = 2
    g = torch.Generator().manual_seed(5)
    x = (0.5 * torch.randn(B, 32, 32, 4, generator=g)).cuda()
    model = VGG16Model(B, x, 6, fc_dim=128, seed=2)
    model.build_graph()
    assert "conv3_2/biases" in model.params and "conv3_2/bn" not in model.params
    model.add_loss(torch.eye(6).cuda(), torch.zeros(B, 6).cuda())
    step = model.add_optimizer()
    model._optimizer.learning_rate = 1e-4
    # first-layer check against float32: conv + bias + relu
    model.forward()
    w, b = model.params["conv1_1/weights"].cpu(), model.params["conv1_1/biases"].cpu()
    xq = x.cpu().to(torch.bfloat16).float().permute(0, 3, 1, 2)
    want = torch.relu(F.conv2d(xq, w.to(torch.bfloat16).float().permute(3, 2, 0, 1), padding=1).to(torch.bfloat16).float() + b.view(1, -1, 1, 1))
    assert rel_l2(model._act[0]["y"].permute(0, 3, 1, 2), want) < 5e-3
    losses = [float(step()) for _ in range(6)]
    assert np.all(np.isfinite(losses)) and losses[-1] < losses[0], losses


def _fresh_model(seed=1):
    from slam_amd.network.cnn_model import VGG16Model
    B, H, W = 4, 64, 64
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, H, W, 4, generator=g).to(torch.bfloat16).float().cuda()
    model = VGG16Model(B, x, 6, 1e-3, fc_dim=256, seed=seed)
    model.build_graph()
    model.add_loss(torch.eye(6).cuda(), torch.randn(B, 6, generator=g).cuda())
    return model


def test_graph_replayed_step_equals_eager_step():
    """capture_train_step(): the step recorded in a HIP graph does what the eager step does from the same
    state (same launches; the staircase learning rate reaches the replayed Adam kernel through device
    memory).  Compared step by step from a restored state: the randomly initialised network amplifies the
    last-bit differences of float atomics over several steps, so whole trajectories are not comparable."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    m = _fresh_model()
    m.add_optimizer(optimizer='AdamOptimizer')
    m._optimizer.learning_rate = lambda s: 1e-4 * 0.1 ** (s // 5)     # the schedule changes inside the run
    step_g = m.capture_train_step(warmup=2)
    assert m.global_step == 2
    deltas = {}
    for start in (2, 4, 5):       # the update of step 6 (global_step 5) uses a ten times smaller rate
        m.global_step = start
        state = [t.clone() for t in (m.flat_params, m._adam_m, m._adam_v)]

        def restore():
            for t, c in zip((m.flat_params, m._adam_m, m._adam_v), state):
                t.copy_(c)
            m.global_step = start
            m.parameters_changed()

        loss_g = float(step_g())
        p_g = m.flat_params.clone()
        assert m.global_step == start + 1
        restore()
        loss_e = float(m.train_step())
        p_e = m.flat_params.clone()
        # (two forward passes from one state differ by the order of the float atomics in the channel
        # statistics, which this small random network amplifies to about a percent of the loss)
        assert abs(loss_g - loss_e) <= 3e-2 * abs(loss_e), (start, loss_g, loss_e)
        d_g, d_e = (p_g - state[0]).double(), (p_e - state[0]).double()
        # (the same noise now and then flips a rounding early in the network, which moves the gradient of this
        # small network by a third; the update kernel itself is compared bit for bit below)
        assert float((d_g - d_e).norm() / d_e.norm()) < 0.6, start
        deltas[start] = float(d_g.norm())
        restore()
    assert 5.0 < deltas[4] / deltas[5] < 20.0, deltas     # the rate fell by ten between the two updates


def test_adam_step_from_device_memory_equals_adam_step():
    """The Adam kernel of the recorded step (step size read from device memory) against the eager one
    (step size as a kernel argument) on the same operands: bit-identical parameters, moments and shadow."""
    from slam_amd.network import encoder_ops as ops
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(4)
    n = 64 * 1000 + 64
    p0, gr = torch.randn(n, generator=g).to(dev), torch.randn(n, generator=g).to(dev)
    m0, v0 = (0.1 * torch.randn(n, generator=g)).to(dev), torch.rand(n, generator=g).to(dev)
    for step, lr in ((1, 1e-2), (7, 1e-3)):
        a = [t.clone() for t in (p0, m0, v0)] + [torch.zeros(4096, dtype=torch.bfloat16, device=dev)]
        b = [t.clone() for t in (p0, m0, v0)] + [torch.zeros(4096, dtype=torch.bfloat16, device=dev)]
        ops.adam_step(a[0], gr, a[1], a[2], lr, step=step, grad_scale=0.5, shadow=a[3])
        lr_t = torch.tensor([ops.adam_lr_t(lr, step)], dtype=torch.float32, device=dev)
        ops.adam_step_dev(b[0], gr, b[1], b[2], lr_t, grad_scale=0.5, shadow=b[3])
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        assert not torch.equal(a[0], p0)


def test_sgd_and_rmsprop_follow_the_tensorflow_formulas():
    """slam/network/cnn_model.py:167-170: GradientDescentOptimizer p -= lr g; RMSPropOptimizer with
    TensorFlow's defaults (decay 0.9, momentum 0, epsilon 1e-10 inside the root, mean square starting at
    one): ms = 0.9 ms + 0.1 g^2, p -= lr g / sqrt(ms + 1e-10).  Three updates each, against float64."""
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slam_amd.network import encoder_ops as ops
    g = torch.Generator().manual_seed(11)
    n = 100003
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * s for s in (1.0, 1e-3, 30.0)]
    lr, scale = 0.01, 0.5
    # SGD
    p = p0.clone().cuda()
    want = p0.double()
    for gr in grads:
        ops.sgd_step(p, gr.cuda(), lr, grad_scale=scale)
        want = want - lr * (gr.double() * scale)
    np.testing.assert_allclose(p.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    # RMSProp
    p = p0.clone().cuda()
    ms = torch.ones(n).cuda()
    want, wms = p0.double(), torch.ones(n, dtype=torch.float64)
    for gr in grads:
        ops.rmsprop_step(p, gr.cuda(), ms, lr, grad_scale=scale)
        gs = gr.double() * scale
        wms = 0.9 * wms + 0.1 * gs * gs
        want = want - lr * gs / torch.sqrt(wms + 1e-10)
    np.testing.assert_allclose(ms.cpu().numpy(), wms.numpy(), rtol=1e-5)
    np.testing.assert_allclose(p.cpu().numpy(), want.numpy(), rtol=1e-5, atol=1e-6)
    # and through the model: the configuration's optimizer names select them
    for kind in ('GradientDescentOptimizer', 'RMSPropOptimizer'):
        m = _fresh_model(seed=4)
        step = m.add_optimizer(optimizer=kind)
        m._optimizer.learning_rate = 1e-4
        before = m.flat_params.clone()
        m.forward(); m.compute_loss(); m.backward()
        grad = m.flat_grads.clone()
        m._apply_gradients(1.0)
        if kind == 'GradientDescentOptimizer':
            want = before.double() - 1e-4 * grad.double()
        else:
            want = before.double() - 1e-4 * grad.double() / torch.sqrt(0.9 + 0.1 * grad.double() ** 2 + 1e-10)
        np.testing.assert_allclose(m.flat_params.cpu().numpy(), want.cpu().numpy(), rtol=1e-5, atol=1e-7)


def test_add_conv_layer_and_add_fc_layer_apply_the_reference_layers():
    """slam/network/cnn_model.py:87-124: relu(batch_norm(conv)) with batch statistics, scale 1 and beta 0;
    relu(x w + b).  Checked against float32 PyTorch on the same bfloat16-rounded operands."""
    from slam_amd.network.cnn_model import VGG16Model
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 64, 64, 4, generator=g).to(dev)
    model = VGG16Model(2, x, 6, 1e-3, fc_dim=256, seed=2)
    model.build_graph()
    before = model.total_weights
    for cin, inp in ((4, x), (64, torch.randn(2, 32, 32, 64, generator=g).to(dev))):
        y = model.add_conv_layer("extra%d" % cin, inp, [3, 3], cin, 128)
        assert y.shape == inp.shape[:3] + (128,) and y.dtype == torch.bfloat16
        w = model.extra_params["extra%d/weights" % cin]
        xb = inp.to(torch.bfloat16).float().permute(0, 3, 1, 2)
        wb = (w.to(torch.bfloat16).float() if cin % 64 == 0 else w).permute(3, 2, 0, 1)
        z = F.conv2d(xb, wb, padding=1).to(torch.bfloat16).float()
        mean, var = z.mean(dim=(0, 2, 3), keepdim=True), z.var(dim=(0, 2, 3), unbiased=False, keepdim=True)
        want = torch.relu((z - mean) / torch.sqrt(var + 1e-3)).permute(0, 2, 3, 1)
        assert rel_l2(y.float(), want.cpu()) < 1e-2
    assert model.total_weights == before + 9 * 4 * 128 + 9 * 64 * 128
    h = torch.randn(8, 128, generator=g).to(dev)
    out = model.add_fc_layer("extra_fc", h, 128, 64)
    w, b = model.extra_params["extra_fc/weights"], model.extra_params["extra_fc/bias"]
    want = torch.relu(h.to(torch.bfloat16).float() @ w.to(torch.bfloat16).float() + b)
    assert out.dtype == torch.float32 and rel_l2(out, want.cpu()) < 5e-3
    with pytest.raises(ValueError):
        model.add_conv_layer("bad", x, [5, 5], 4, 64)


def test_graph_replays_stay_finite_with_host_reads_between_steps():
    """Replays of the recorded step interleaved with allocations and device-to-host reads (what a training
    loop's logging does) follow the eager trajectory.  (With memset nodes in the recorded graph such
    replays now and then read accumulators that had not been cleared; the library clears with kernels.)"""
    from slam_amd.network.cnn_model import VGG16Model
    dev = torch.device("cuda:0")

    def run(graph):
        torch.manual_seed(0)
        B = 32
        x = torch.randn(B, 224, 224, 4, device=dev)
        model = VGG16Model(B, x, 6, 1e-3, fc_dim=1024)
        model.build_graph()
        model.add_loss(torch.eye(6, device=dev), torch.randn(B, 6, device=dev) * 0.1)
        step = model.add_optimizer(optimizer='AdamOptimizer')
        model._optimizer.learning_rate = 1e-5
        if graph:
            step = model.capture_train_step(warmup=2)
        else:
            step(), step()
        losses = []
        for _ in range(6):
            step()
            for t in list(model.grads.values()) + list(model.params.values()):
                assert bool(torch.isfinite(t).all())
            losses.append(float(model.loss))
        return np.array(losses)

    eager, replayed = run(False), run(True)
    assert np.isfinite(replayed).all()
    # The two trajectories are compared bit for bit under fixed-order sums in
    # test_encoder_fullsize_gpu.py::test_graph_replays_back_to_back_equal_eager_steps_bit_for_bit; with the default
    # float atomics two runs of ONE path already spread by 1 % at the first step and 12 % and more from the third
    # (a chaotic trajectory at bfloat16 resolution), so that here only the first steps are held together, loosely.
    np.testing.assert_allclose(replayed[:2], eager[:2], rtol=0.25)
    assert (replayed > 0).all() and replayed.max() < 10 * eager.max()
