"""The frame-pair, overlap and input-preparation entry points recorded in a HIP graph and replayed on new
inputs: they launch kernels only (their clears are kernels too), so a replay must give what an eager call on the
same inputs gives."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from slam_amd import capi
from slam_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _host_inputs(B, H, W, seed):
    from oracle import slam_oracle as so
    gt = syn.pose_pairs(B, seed=seed, max_deg=5.0, max_t=0.3)
    rel = np.stack([so.relative_pose(so.quat_to_transformation(gt[i, 0]), so.quat_to_transformation(gt[i, 1]))
                    for i in range(B)]).reshape(B, 1, 4, 4)
    return [syn.depth_frames(B, H, W, seed=seed, kind="noise"), syn.rgb_frames(B, H, W, seed=seed, kind="noise"),
            syn.depth_frames(B, H, W, seed=seed + 1, kind="noise").reshape(B, 1, H, W),
            syn.rgb_frames(B, H, W, seed=seed + 1, kind="noise").reshape(B, 1, H, W, 3), rel]


def test_replays_follow_eager_calls():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from slam_amd import frame_pairs as fp
    from slam_amd.network import prep
    lib = capi.load()
    B, H, W = 5, 96, 256
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    dev = torch.device("cuda")
    t = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in _host_inputs(B, H, W, seed=31)]
    sums = torch.empty((B, 1, capi.NSUMS), dtype=torch.float64, device=dev)
    counts = torch.empty((B, 1, capi.NCOUNTS), dtype=torch.int32, device=dev)
    ovl = torch.empty((B, 2), dtype=torch.int32, device=dev)
    nws = lib.slamhip_prep_rgbd_workspace_bytes(B)
    prep_ws = torch.empty((nws,), dtype=torch.uint8, device=dev)
    prep_out = torch.empty((B, 32, 48, 4), dtype=torch.float32, device=dev)
    prep_mean = torch.empty((4,), dtype=torch.float64, device=dev)

    def calls(stream):
        head = [x.data_ptr() for x in t] + [capi.K4(K), 5000.0, B, 1, H, W, sums.data_ptr(), counts.data_ptr()]
        capi.check(lib.slamhip_pair_loss_fused_static(*head, stream), "slamhip_pair_loss_fused_static")
        capi.check(lib.slamhip_overlap_fused(t[0].data_ptr(), t[4].data_ptr(), capi.K4(K), 5000.0, float(W), float(H), B, H, W,
                                             ovl.data_ptr(), stream), "slamhip_overlap_fused")
        capi.check(lib.slamhip_prep_rgbd(t[1].data_ptr(), t[0].data_ptr(), B, H, W, 32, 48, prep_out.data_ptr(), None,
                                         prep_mean.data_ptr(), prep_ws.data_ptr(), nws, stream), "slamhip_prep_rgbd")

    def snapshot():
        torch.cuda.synchronize()
        return [x.clone() for x in (sums, counts, ovl, prep_out, prep_mean)]

    calls(capi.current_stream())      # eager once: the static entry looks its array up outside a capture
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        calls(capi.current_stream())
    for seed in (41, 51):
        for dst, src in zip(t, _host_inputs(B, H, W, seed=seed)):
            dst.copy_(torch.from_numpy(np.ascontiguousarray(src)))
        calls(capi.current_stream())
        want = snapshot()
        for x in (sums, counts, ovl, prep_out, prep_mean):
            x.zero_()
        torch.cuda.synchronize()
        g.replay()
        got = snapshot()
        assert torch.equal(got[1], want[1]) and torch.equal(got[2], want[2])
        assert int(got[1][..., 0].min()) > 0 and int(got[2][:, 1].min()) > 0
        np.testing.assert_allclose(got[0].cpu().numpy(), want[0].cpu().numpy(), rtol=1e-9)
        np.testing.assert_allclose(got[3].cpu().numpy(), want[3].cpu().numpy(), rtol=1e-6, atol=1e-5)
        np.testing.assert_allclose(got[4].cpu().numpy(), want[4].cpu().numpy(), rtol=1e-12)


def test_google_net_layers_replay_like_eager_calls():
    """Forward and backward of a small graph that uses every GoogLeNet layer kernel (strided 7x7, 5x5, 1x1, concat
    slices, fan-out sums, SAME / VALID pools, LRN, fc heads; slam/network/google_net.py:14-215) recorded in a HIP
    graph and replayed on new inputs: outputs and every parameter gradient as the eager calls give them (the
    library clears the bias-gradient and filter-gradient accumulators with kernels: the recording holds kernel
    nodes only)."""
    assert torch.cuda.is_available(), "GPU tests need a GPU"
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
            (self.feed('cat').max_pool(3, 3, 2, 2, name='pool2').avg_pool(3, 3, 1, 1, padding='VALID', name='gap')
             .fc(24, relu=False, name='fc1').fc(7, relu=False, name='out_main'))

    dev = torch.device("cuda")
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 45, 53, 3, generator=g).to(dev)
    net = SmallNet({'data': x}, seed=1)
    d_out = net.layers['out_main'].d_out

    def body():
        net.forward()
        net.backward()

    d_out.copy_(torch.randn(3, 7, generator=g).to(dev))
    body()                       # eager once: every lazily made buffer exists before the recording
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=side):
        body()
    for seed in (6, 7):
        gg = torch.Generator().manual_seed(seed)
        x.copy_(torch.randn(3, 45, 53, 3, generator=gg).to(dev))
        d_out.copy_(torch.randn(3, 7, generator=gg).to(dev))
        body()
        torch.cuda.synchronize()
        want_out = net.outputs()['out_main'].clone()
        want_g = net.flat_grads.clone()
        net.flat_grads.fill_(7.0)
        torch.cuda.synchronize()
        graph.replay()
        torch.cuda.synchronize()
        assert float(want_g.abs().max()) > 0
        np.testing.assert_allclose(net.outputs()['out_main'].cpu().numpy(), want_out.cpu().numpy(), rtol=1e-5, atol=1e-6)
        scale = float(want_g.abs().max())
        np.testing.assert_allclose(net.flat_grads.cpu().numpy(), want_g.cpu().numpy(), rtol=1e-3, atol=1e-4 * scale)
