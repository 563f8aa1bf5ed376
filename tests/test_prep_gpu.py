"""GPU input preparation against the oracle's restatement of the same steps (oracle/prep_oracle.py: integer
depth scaling, order-1 resize, batch mean; parity unpinned, skimage and TensorFlow are not installed)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,Ho,Wo", [(3, 480, 640, 224, 224), (2, 37, 53, 16, 24), (1, 20, 20, 20, 20)])
def test_rgbd_batch_matches_host(B, H, W, Ho, Wo):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from oracle import prep_oracle, slam_oracle as so
    from slam_amd.network import prep
    from slam_amd.utils import synthetic as syn
    rgb = syn.rgb_frames(B, H, W, seed=3, kind="smooth")
    depth = syn.depth_frames(B, H, W, seed=4, kind="plane")
    depth[-1] = 0                                    # an all-zero depth frame: max clamps to 1
    imgs = [prep_oracle.get_rgbd_file(rgb[b], depth[b], Ho, Wo) for b in range(B)]
    mean = so.get_mean(imgs, channels=4)
    want = np.stack([(im - mean).astype(np.float32) for im in imgs])
    out_b = torch.empty(B, Ho, Wo, 4, dtype=torch.bfloat16, device="cuda")
    out, m = prep.rgbd_batch(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), (Ho, Wo), out_bf16=out_b)
    np.testing.assert_allclose(m.cpu().numpy(), mean, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=2e-4)
    assert torch.equal(out_b, out.to(torch.bfloat16))


@pytest.mark.parametrize("B,H,W,Ho,Wo", [(2, 480, 640, 224, 224), (2, 37, 53, 16, 24), (1, 20, 20, 20, 20)])
def test_in_graph_reader_variant_matches_restatement(B, H, W, Ho, Wo):
    """slam/network/model_input.py:185-217: 8-bit depth, TensorFlow bilinear, depth * 255 / max."""
    from oracle import prep_oracle
    from slam_amd.network import prep
    from slam_amd.utils import synthetic as syn
    rgb = syn.rgb_frames(B, H, W, seed=5, kind="noise")
    depth = syn.depth_frames(B, H, W, seed=6, kind="noise")
    want = np.stack([prep_oracle.read_rgbd_data(rgb[b], depth[b], Ho, Wo) for b in range(B)])
    out = prep.rgbd_batch_tf(torch.from_numpy(rgb).cuda(), torch.from_numpy(depth).cuda(), (Ho, Wo))
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-5, atol=1e-4)
    assert abs(float(out[..., 3].amax()) - 255.0) < 1e-3


def test_queued_input_provider(tmp_path, monkeypatch):
    """model_input.py:126-319: per-sequence cyclic queues, in-graph reader arithmetic, twists of
    model_input's _get_data (relative to the first frame, clamped variant)."""
    import json
    import os
    from oracle import prep_oracle, slam_oracle as so
    from slam_amd.network import model_config as mc
    from slam_amd.network import model_input as mi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = json.load(open(os.path.join(root, "resources", "model.config")))
    cfg["train"]["slam_files"] = ["freiburg1_xyz", "freiburg1_rpy"]
    p = tmp_path / "model.config"
    p.write_text(json.dumps(cfg))
    provider_cfg = mc.ModelConfigProvider(str(p))
    for i, name in enumerate(provider_cfg.training_filenames()):
        mi.write_synthetic_sequence(str(tmp_path / "tum" / name), n_frames=3, height=48, width=64, seed=i)
    q = mi.QueuedInputProvider(str(tmp_path / "tum"), config_provider=provider_cfg)
    assert q.get_batch_size() == 2
    images, twists = q.get_training_batch()
    assert len(images) == 2 and images[0].shape == (224, 224, 4) and twists[0].shape == (1, 1, 6)
    seq = str(tmp_path / "tum" / "freiburg1_xyz")
    assoc = np.loadtxt(os.path.join(seq, "associate.txt"), dtype="str")
    gt = np.loadtxt(os.path.join(seq, "groundtruth.txt"), dtype="str")
    T0 = so.quat_to_transformation(gt[0, 1:].astype(np.float32))
    for k in range(4):                                   # 3 frames, then the queue wraps around
        i = k % 3
        want = prep_oracle.read_rgbd_data(mi.read_png(os.path.join(seq, assoc[i, 1])),
                                          mi.read_png(os.path.join(seq, assoc[i, 3])).astype(np.uint16))
        np.testing.assert_allclose(images[0].cpu().numpy(), want, rtol=1e-5, atol=1e-4)
        tw = np.zeros(6) if i == 0 else np.asarray(so.trans_to_twist(
            so.relative_pose(T0, so.quat_to_transformation(gt[i, 1:].astype(np.float32))), clamp=True)).ravel()
        np.testing.assert_allclose(twists[0].cpu().numpy().ravel(), tw, rtol=1e-5, atol=1e-7)
        q.run()
