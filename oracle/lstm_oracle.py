"""TEST INFRASTRUCTURE - CPU restatement (torch float64, differentiable) of the
recurrent head of the reference's CNN-LSTM trainer: slam/network/lstm_model.py:25-55.

The cell arithmetic lives in TensorFlow 0.x (`tensorflow.python.ops.rnn_cell.LSTMCell`
and `MultiRNNCell`, un-vendored and unpinned in the reference), restated from its
published definition: no peepholes, no projection, forget_bias 1.0, gate order
(i, j, f, o), layer state [c, m], stack state = layer states side by side.
Parity unpinned: TensorFlow is not installed and the reference holds no fixture
for this path.
"""
import torch


def lstm_head(x, state, W, bias, fc_w, fc_b, forget_bias=1.0):
    """x [B,n], state [B,2nL], W [L,2n,4n], bias [L,4n], fc_w [n,D], fc_b [D]
    -> (out [B,D], new state [B,2nL])."""
    L, n = W.shape[0], x.shape[1]
    h, new_state = x, []
    for l in range(L):
        c_prev = state[:, l * 2 * n:l * 2 * n + n]
        m_prev = state[:, l * 2 * n + n:(l + 1) * 2 * n]
        i, j, f, o = torch.split(torch.cat([h, m_prev], 1) @ W[l] + bias[l], n, dim=1)
        c = torch.sigmoid(f + forget_bias) * c_prev + torch.sigmoid(i) * torch.tanh(j)
        m = torch.sigmoid(o) * torch.tanh(c)
        new_state += [c, m]
        h = m
    return h @ fc_w + fc_b, torch.cat(new_state, 1)


def lstm_loss(out, gt, loss_weight):
    """lstm_model.py:52-55"""
    return (((out - gt) @ loss_weight) ** 2).sum() / out.shape[0]
