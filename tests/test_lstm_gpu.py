"""Recurrent head (stacked LSTM cells + output layer) on the GPU against the float64
restatement: forward values, every gradient (autograd of the restatement), the
trainer-facing class, and the gradient that flows into the encoder."""
import numpy as np
import pytest
import torch

from oracle import lstm_oracle
from slam_amd.network import encoder_ops as ops

pytestmark = pytest.mark.gpu


def _case(B, n, L, D, seed, zero_state=False):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    t = dict(x=r(B, n), state=torch.zeros(B, 2 * n * L, dtype=torch.float64) if zero_state else 0.5 * r(B, 2 * n * L),
             W=0.5 * r(L, 2 * n, 4 * n), bias=0.1 * r(L, 4 * n), fc_w=r(n, D), fc_b=r(D), gt=r(B, D), lw=r(D, D))
    return t


@pytest.mark.parametrize("B,n,L,D,zero_state", [(1, 6, 5, 6, True), (4, 6, 5, 6, False), (3, 7, 2, 7, False),
                                                (130, 64, 3, 64, False), (2, 16, 1, 3, False)])
def test_lstm_head_forward_and_gradients(B, n, L, D, zero_state):
    t = _case(B, n, L, D, seed=B + n, zero_state=zero_state)
    leaves = {k: t[k].clone().requires_grad_(True) for k in ("x", "W", "bias", "fc_w", "fc_b")}
    out, new_state = lstm_oracle.lstm_head(leaves["x"], t["state"], leaves["W"], leaves["bias"], leaves["fc_w"],
                                           leaves["fc_b"])
    out.retain_grad()
    loss = lstm_oracle.lstm_loss(out, t["gt"], t["lw"])
    loss.backward()

    dev = torch.device("cuda")
    d = {k: v.to(dev, torch.float32).contiguous() for k, v in t.items()}
    f32 = dict(dtype=torch.float32, device=dev)
    gates, state_out = torch.empty(B, L, 4 * n, **f32), torch.empty(B, 2 * n * L, **f32)
    got = torch.empty(B, D, **f32)
    ops.lstm_head_fwd(d["x"], d["state"], d["W"], d["bias"], d["fc_w"], d["fc_b"], gates, state_out, got)
    np.testing.assert_allclose(got.cpu().numpy(), out.detach().numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(state_out.cpu().numpy(), new_state.detach().numpy(), rtol=1e-4, atol=1e-5)

    if D <= 8:      # the loss kernel covers the trainers' widths; wider heads take the restatement's gradient
        loss_gpu, d_out = torch.empty(1, **f32), torch.empty(B, D, **f32)
        ops.weighted_sq_loss(got, d["gt"], d["lw"], 1.0 / B, loss_gpu, d_out)
        np.testing.assert_allclose(float(loss_gpu), float(loss.detach()), rtol=1e-4)
    else:
        d_out = out.grad.to(dev, torch.float32).contiguous()
    d_pre, d_x = torch.empty(B, L, 4 * n, **f32), torch.empty(B, n, **f32)
    grads = {k: torch.empty_like(d[k]) for k in ("W", "bias", "fc_w", "fc_b")}
    ops.lstm_head_bwd(d_out, d["x"], d["state"], d["W"], d["fc_w"], gates, state_out, d_pre, d_x,
                      grads["W"], grads["bias"], grads["fc_w"], grads["fc_b"])
    grads["x"] = d_x
    for k, gpu in grads.items():
        want = leaves[k].grad.numpy()
        scale = max(np.abs(want).max(), 1e-6)
        np.testing.assert_allclose(gpu.cpu().numpy() / scale, want / scale, rtol=0, atol=2e-4, err_msg=k)


def test_lstm_model_trains_and_ignores_fed_state():
    from slam_amd.network.lstm_model import LSTMmodel
    dev = torch.device("cuda")
    B = 4
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 6, generator=g).to(dev)
    gt = (0.1 * torch.randn(B, 6, generator=g)).to(dev)
    fed_state = torch.ones(B, 2 * 6 * 5, device=dev)              # has no effect, as in the reference
    m = LSTMmodel(x, layer_size=6, layers=5, output_dim=6, ground_truth=gt, batch_size=B, init_state=fed_state)
    out = m.build_graph()
    assert out.shape == (B, 6) and m.init_state.shape == (B, 60)
    loss = m.add_loss(torch.eye(6, device=dev))
    step = m.add_optimizer()
    # the forward pass equals the restatement with a zero state
    m.forward()
    p = {k: v.double().cpu() for k, v in m.params.items()}
    want, want_state = lstm_oracle.lstm_head(x.double().cpu(), torch.zeros(B, 60, dtype=torch.float64),
                                             p["lstm/MultiRNNCell/W"], p["lstm/MultiRNNCell/B"],
                                             p["lstm/fc_out/weights"], p["lstm/fc_out/biases"])
    np.testing.assert_allclose(out.cpu().numpy(), want.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(m.init_state.cpu().numpy(), want_state.numpy(), rtol=1e-4, atol=1e-5)
    assert float(p["lstm/MultiRNNCell/W"].abs().max()) <= 0.5 and float(p["lstm/MultiRNNCell/B"].abs().max()) == 0.0
    losses = [float(step()) for _ in range(40)]
    assert losses[-1] < 0.5 * losses[0]
    assert m.global_step == 40
    assert m.learning_rate(0) == 0.01 and abs(m.learning_rate(50) - 0.001) < 1e-12 and m.learning_rate(49) == 0.01


def test_cnn_lstm_network_step_updates_encoder_and_head():
    """main/lstm_rgbd.py:71-77 - the head's gradient reaches the encoder's output layer
    exactly (d out_layer/bias = column sums of d_x) and both parameter sets move."""
    from slam_amd.main.lstm_rgbd import build_complete_network
    dev = torch.device("cuda")
    B = 2
    g = torch.Generator().manual_seed(5)
    rgbd = torch.randn(B, 64, 64, 4, generator=g).to(dev)
    gt = (0.1 * torch.randn(B, 6, generator=g)).to(dev)
    state = torch.zeros(B, 60, device=dev)
    cnn, cnn_out, lstm, lstm_out = build_complete_network(rgbd, gt, state, B, 5, 6, 1e-3, fc_dim=256)
    loss = lstm.add_loss(torch.eye(6, device=dev))
    step = lstm.add_optimizer()
    lstm.forward()
    lstm.compute_loss()
    lstm.backward()
    np.testing.assert_allclose(cnn.grads["out_layer/bias"][:6].cpu().numpy(), lstm._d_x.sum(0).cpu().numpy(),
                               rtol=1e-5, atol=1e-7)
    assert float(cnn.grads["conv1_1/weights"].abs().max()) > 0
    before_cnn, before_head = cnn.flat_params.clone(), lstm.flat_params.clone()
    first = float(step())
    assert np.isfinite(first)
    assert not torch.equal(before_cnn, cnn.flat_params) and not torch.equal(before_head, lstm.flat_params)
    assert set(lstm.named_variables()) >= {"lstm/fc_out/weights", "conv1_1/weights", "out_layer/bias"}
