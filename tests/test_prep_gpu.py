"""GPU input preparation against the host implementation of the same steps
(slam_amd/network/model_input.py: integer depth scaling, bilinear resize, get_mean)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,Ho,Wo", [(3, 480, 640, 224, 224), (2, 37, 53, 16, 24), (1, 20, 20, 20, 20)])
def test_rgbd_batch_matches_host(B, H, W, Ho, Wo):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slam_amd.network import model_input as mi
    from slam_amd.network import prep
    from slam_amd.utils import synthetic as syn
    rgb = syn.rgb_frames(B, H, W, seed=3, kind="smooth")
    depth = syn.depth_frames(B, H, W, seed=4, kind="plane")
    depth[-1] = 0                                    # an all-zero depth frame: max clamps to 1
    imgs = []
    for b in range(B):
        d = depth[b].astype(np.int64)[..., None]
        d = 255 * d // max(int(d.max()), 1)
        imgs.append(mi.resize_bilinear(np.concatenate((rgb[b], d), 2), Ho, Wo).astype(np.float32))
    mean = mi.get_mean(imgs, channels=4)
    want = np.stack([(im - mean).astype(np.float32) for im in imgs])
    out_b = torch.empty(B, Ho, Wo, 4, dtype=torch.bfloat16, device="cuda")
    out, m = prep.rgbd_batch(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), (Ho, Wo), out_bf16=out_b)
    np.testing.assert_allclose(m.cpu().numpy(), mean, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-4)
    assert torch.equal(out_b, out.to(torch.bfloat16))
