"""GoogLeNet encoder: every layer kernel through a small graph that uses all of them
(strided 7x7, 5x5, 1x1, concat slices, fan-out, SAME / VALID pools, LRN, fc heads)
against the PyTorch restatement, then the full network."""
import numpy as np
import pytest
import torch

from oracle import torch_network

pytestmark = pytest.mark.gpu


def _rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-12)


def _small_net_class():
    from slam_amd.network.network import Network

    class SmallNet(Network):
        def setup(self):
            (self.feed('data').conv(7, 7, 16, 2, 2, name='stem').max_pool(3, 3, 2, 2, name='pool1')
             .lrn(2, 2e-05, 0.75, name='norm1').conv(1, 1, 24, 1, 1, name='reduce'))
            self.feed('reduce').conv(1, 1, 8, 1, 1, name='b1')
            self.feed('reduce').conv(1, 1, 16, 1, 1, name='b3r').conv(3, 3, 24, 1, 1, name='b3')
            self.feed('reduce').conv(1, 1, 8, 1, 1, name='b5r').conv(5, 5, 16, 1, 1, name='b5')
            self.feed('reduce').max_pool(3, 3, 1, 1, name='bp').conv(1, 1, 16, 1, 1, name='bpp')
            self.feed('b1', 'b3', 'b5', 'bpp').concat(3, name='cat')
            (self.feed('cat').avg_pool(5, 5, 3, 3, name='aux_pool').conv(1, 1, 64, 1, 1, name='aux_conv')
             .fc(32, name='aux_fc').fc(7, name='out_aux', relu=False))
            (self.feed('cat').max_pool(3, 3, 2, 2, name='pool2').avg_pool(3, 3, 1, 1, padding='VALID', name='gap')
             .fc(24, relu=False, name='fc1').fc(7, relu=False, name='out_main'))
    return SmallNet


def _oracle_pass(net, x, head_grads):
    variables = {k: v.detach().float().cpu().clone().contiguous().requires_grad_(True)
                 for k, v in net.named_variables().items()}
    acts = torch_network.run(net.spec, variables, x.cpu())
    loss = sum((acts[name].reshape(g.shape[0], -1) * g.cpu()).sum() for name, g in head_grads.items())
    loss.backward()
    return acts, variables


def test_small_graph_forward_and_gradients():
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(3, 45, 53, 3, generator=g).to(dev)
    net = _small_net_class()({'data': x}, seed=1)
    assert net.layers['stem'].shape == (3, 23, 27, 16) and net.layers['pool1'].shape == (3, 12, 14, 16)
    assert net.layers['cat'].shape == (3, 12, 14, 64) and net.layers['aux_pool'].shape == (3, 4, 5, 64)
    assert net.layers['gap'].shape == (3, 4, 5, 64) and net.layers['out_main'].shape == (3, 1, 1, 7)
    net.forward()
    head_grads = {}
    for name, node in (('out_aux', net.layers['out_aux']), ('out_main', net.layers['out_main'])):
        node.d_out.copy_(torch.randn(3, 7, generator=g).to(dev))
        head_grads[name] = node.d_out.clone()
    net.backward()
    acts, variables = _oracle_pass(net, x, head_grads)
    for name in ('stem', 'pool1', 'norm1', 'reduce', 'cat', 'aux_pool', 'pool2', 'gap'):
        node = net.layers[name]
        assert node.whole()
        assert _rel(node.base.float().cpu().numpy(), acts[name].detach().numpy()) < 1e-2, name
    for name in ('out_aux', 'out_main'):
        assert _rel(net.outputs()[name].cpu().numpy(), acts[name].detach().reshape(3, 7).numpy()) < 1e-2, name
    grads = net.named_gradients()
    for name, v in variables.items():
        assert _rel(grads[name].cpu().numpy(), v.grad.numpy()) < 3e-2, name
    # the padding rows / columns of the filter matrices and the unused input channels stay zero
    assert float(net.flat_grads.abs().sum()) == pytest.approx(
        float(sum(g.abs().sum() for g in grads.values())), rel=1e-5)


@pytest.mark.parametrize("lrn", [True, False])
def test_google_net_forward_matches_restatement_and_trains(lrn):
    from slam_amd.network import google_net, google_net_noLRN
    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(2)
    B = 2
    x = torch.randn(B, 224, 224, 3, generator=g).to(dev)
    gt = torch.randn(B, 7, generator=g).to(dev)
    cls = google_net.GoogleNet if lrn else google_net_noLRN.GoogleNet
    net = cls({'data': x}, 7)
    assert ('pool1_norm1' in net.layers) == lrn
    assert net.layers['inception_3a_output'].shape == (B, 28, 28, 256)
    assert net.layers['inception_4e_output'].shape == (B, 14, 14, 832)
    assert net.layers['inception_5b_output'].shape == (B, 7, 7, 1024)
    assert net.layers['inception_4b_avg_pool'].shape == (B, 5, 5, 512)
    assert sorted(net.outputs()) == ['output1', 'output2', 'output3']
    net.forward()
    variables = {k: v.detach().float().cpu().clone().contiguous() for k, v in net.named_variables().items()}
    with torch.no_grad():
        acts = torch_network.run(net.spec, variables, x.cpu())
    for name in ('conv1_7x7_s2', 'pool2_3x3_s2', 'inception_3b_output', 'inception_4e_output', 'inception_5b_output'):
        assert _rel(net.layers[name].base.float().cpu().numpy(), acts[name].numpy()) < 2e-2, name
    for name in ('output1', 'output2', 'output3'):
        assert _rel(net.outputs()[name].cpu().numpy(), acts[name].reshape(B, 7).numpy()) < 2e-2, name
    # loss (google_net.py:217-225) and one optimizer step
    total = net.add_loss(gt)
    step = net.add_optimizer()
    net.compute_loss()
    want = 0.0
    for name, w in zip(('output1', 'output2', 'output3'), (0.3, 0.3, 1.0)):
        o, t = acts[name].reshape(B, 7).double().numpy(), gt.double().cpu().numpy()
        want += w * (((o[:, :3] - t[:, :3]) ** 2).sum() + 1100 * ((o[:, 3:] - t[:, 3:]) ** 2).sum())
    assert abs(float(total) - want) / want < 3e-2
    before = net.flat_params.clone()
    loss0 = float(step())
    assert np.isfinite(loss0) and net.global_step == 1
    assert not torch.equal(before, net.flat_params) and bool(torch.isfinite(net.flat_params).all())
    assert net.learning_rate(0) == 1e-5 and abs(net.learning_rate(10) - 1e-6) < 1e-15
    # the l2 variant (google_net.py:227-240)
    net.add_l2_loss(gt)
    net.forward()
    got = float(net.compute_loss())
    outs = {k: v.double().cpu().numpy() for k, v in net.outputs().items()}
    want = sum(w * 0.5 * ((outs[n] - gt.double().cpu().numpy()) ** 2).sum()
               for n, w in zip(('output1', 'output2', 'output3'), (0.3, 0.3, 1.0)))
    assert abs(got - want) / want < 1e-4
