"""The allocation-free entry point of the fused frame-pair kernel against the
entry point that takes a caller's workspace: counts bit-exact, sums to 1e-9
relative (float64 atomics add in a different order from launch to launch), on two
calls in a row, the second with a smaller input; and against the oracle (counts bit-exact,
sums to 1e-4).  Through the C ABI."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from slam_amd import capi
from slam_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs(B, k, H, W, seed):
    from oracle import slam_oracle as so
    gt = syn.pose_pairs(B * k, seed=seed)
    d1 = syn.depth_frames(B, H, W, seed=seed, kind="plane")
    c1 = syn.rgb_frames(B, H, W, seed=seed, kind="noise")
    d2 = syn.depth_frames(B * k, H, W, seed=seed + 1, kind="plane").reshape(B, k, H, W)
    c2 = syn.rgb_frames(B * k, H, W, seed=seed + 1, kind="noise").reshape(B, k, H, W, 3)
    rel = np.stack([so.relative_pose(so.quat_to_transformation(gt[i, 0]), so.quat_to_transformation(gt[i, 1]))
                    for i in range(B * k)]).reshape(B, k, 4, 4)
    return [_dev(a) for a in (d1, c1, d2, c2, rel)]


def _both(lib, t, B, k, H, W, K, factor):
    dev = t[0].device
    stream = capi.current_stream()
    out = []
    for static in (False, True):
        sums = torch.full((B, k, capi.NSUMS), -1.0, dtype=torch.float64, device=dev)
        counts = torch.full((B, k, capi.NCOUNTS), -1, dtype=torch.int32, device=dev)
        head = [x.data_ptr() for x in t] + [capi.K4(K), float(factor), B, k, H, W, sums.data_ptr(), counts.data_ptr()]
        if static:
            rc = lib.slamhip_pair_loss_fused_static(*head, stream)
            capi.check(rc, "slamhip_pair_loss_fused_static")
        else:
            nbytes = lib.slamhip_pair_loss_workspace_bytes(B, k)
            ws = torch.zeros((nbytes,), dtype=torch.uint8, device=dev)
            rc = lib.slamhip_pair_loss_fused(*head, ws.data_ptr(), nbytes, stream)
            capi.check(rc, "slamhip_pair_loss_fused")
        torch.cuda.synchronize()
        out.append((sums.cpu().numpy(), counts.cpu().numpy()))
    (s0, c0), (s1, c1) = out
    assert np.array_equal(c0, c1)
    assert (c0[..., 0] > 0).all()
    np.testing.assert_allclose(s1, s0, rtol=1e-9)
    # ... and against the oracle itself: counts bit for bit, sums within the float32 tolerance of the path
    from oracle import slam_oracle as so
    want_s, want_c = so.pair_loss(*[x.cpu().numpy() for x in t], K, factor)
    assert np.array_equal(c1.astype(np.int64) & 0xFFFFFFFF, want_c.astype(np.int64))
    np.testing.assert_allclose(s1, want_s, rtol=1e-4, atol=1e-6)


def test_static_entry_matches_workspace_entry():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    lib = capi.load()
    K = (517.3, 516.5, 318.6, 255.3)
    # two calls in a row on the same GPU: the second is smaller and must not see the first one's records
    _both(lib, _inputs(6, 2, 96, 256, seed=11), 6, 2, 96, 256, K, 1.0)
    _both(lib, _inputs(2, 1, 48, 64, seed=12), 2, 1, 48, 64, K, 5000.0)


def test_static_entry_rejects_oversize():
    lib = capi.load()
    p8 = 8
    rc = lib.slamhip_pair_loss_fused_static(p8, p8, p8, p8, p8, capi.K4((1, 1, 0.5, 0.5)), 1.0,
                                            capi.STATIC_MAX_PAIRS + 1, 1, 8, 8, p8, p8, None)
    assert rc == -3 and b"scratch" in lib.slamhip_last_error()
