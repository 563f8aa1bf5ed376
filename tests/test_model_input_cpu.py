"""The input provider on a synthetic TUM-shaped directory (CPU only): labels
against the oracle's restatement of model_input.py:384-396, per-batch mean
subtraction against get_mean (:517-522, golden-pinned in test_oracle_golden),
the Python-2 integer depth scaling, iterator semantics.
"""
import os

import numpy as np
import pytest

from oracle import slam_oracle as so
from slam_amd.network import model_input as mi


@pytest.fixture(scope="module")
def provider(tmp_path_factory):
    base = tmp_path_factory.mktemp("tum")
    names = ["seq_a", "seq_b"]
    for i, n in enumerate(names):
        mi.write_synthetic_sequence(str(base / n), n_frames=5, height=48, width=64, seed=i)
    return mi.SimpleInputProvider(lambda: list(names), base_data_dir=str(base)), base, names


def test_labels_are_absolute_twists(provider):
    p, base, names = provider
    for seq_dir, entry in p.seq_dir_map.items():
        gt = entry["groundtruth"]
        rows = np.array([gt[:, 1:][so.find_label(gt[:, 0], s)] for s in entry["associations"][:, 0]]).astype(np.float32)
        want = so.absolute_twist_labels(rows)
        np.testing.assert_allclose(entry["relpos"], want, rtol=1e-11, atol=1e-13)
        assert not entry["relpos"][0].any() and entry["relpos"][1:].any()


def test_get_mean_matches_oracle():
    rs = np.random.RandomState(0)
    imgs = [rs.uniform(0, 255, (7, 9, 4)) for _ in range(3)]
    np.testing.assert_allclose(mi.get_mean(imgs, channels=4), so.get_mean(imgs, channels=4), rtol=1e-15)


def test_rgbd_frame_and_batches(provider):
    p, base, names = provider
    seq = p.sequence_dirs[0]
    rgb_name, depth_name, img = p.get_rgbd_file(seq, 1)
    assert img.shape == (224, 224, 4) and img.dtype == np.float32
    assert "rgb/" in rgb_name and "depth/" in depth_name
    d = mi.read_png(depth_name)
    assert d.dtype == np.int32 and d.max() > 255
    # depth channel: integer 255*d//max, then interpolation stays inside [0, 255]
    assert 0 <= img[..., 3].min() and img[..., 3].max() <= 255
    scaled = 255 * d.astype(np.int64) // d.max()
    assert abs(float(img[..., 3].mean()) - float(scaled.mean())) < 2.0
    it = p.sequence_batch_itr(3, 4)
    batches = list(it)
    assert len(batches) == 3
    for b in batches:
        assert len(b.rgbd_images) == 4 and len(b.groundtruths) == 4
        # per-batch channel means are removed
        np.testing.assert_allclose(np.mean(np.stack(b.rgbd_images), axis=(0, 1, 2)), 0.0, atol=1e-3)
    full = list(p.complete_seq_iter())
    assert len(full) == 5 and len(full[0].rgbd_images) == 1
    assert not np.any(full[0].groundtruths[0])      # first frame: zero label


def test_resize_is_identity_at_same_size_and_exact_on_ramps():
    x = np.arange(12.0).reshape(3, 4, 1)
    assert np.array_equal(mi.resize_bilinear(x, 3, 4), x)
    ramp = np.tile(np.arange(8.0)[None, :, None], (4, 1, 1))
    out = mi.resize_bilinear(ramp, 4, 4)
    np.testing.assert_allclose(out[0, :, 0], [0.5, 2.5, 4.5, 6.5])


def test_posenet_input_provider(tmp_path):
    """model_input.py:456-515 - rows of dataset_train.txt in order, scale-to-256 + 224 crop,
    per-batch mean subtraction, and the reference's end condition (index + batch < len)."""
    from slam_amd.network import model_input as mi
    d = mi.write_synthetic_posenet_dataset(str(tmp_path / "kings"), n_images=5, height=270, width=480)
    provider = mi.PoseNetInputProvider(d)
    assert provider.sequence_info.shape == (5, 8)                 # the three title lines are skipped
    batches = list(provider.sequence_batch_itr(2))
    assert len(batches) == 2                                      # rows 0-1 and 2-3; 4 + 2 >= 5 ends the pass
    b = batches[0]
    assert [os.path.basename(f) for f in b.rgb_filenames] == ["frame00000.png", "frame00001.png"]
    assert b.groundtruths.shape == (2, 7) and all(img.shape == (224, 224, 3) for img in b.rgb_files)
    np.testing.assert_allclose(mi.get_mean(b.rgb_files), 0.0, atol=1e-9)
    assert len(list(provider.sequence_batch_itr(5))) == 0         # 0 + 5 < 5 is false
    # scaling: the smaller side becomes 256, the other round(side * factor)
    class Fixed:
        @staticmethod
        def randint(lo, hi):
            return hi - 1
    img = mi.read_rgb_image(os.path.join(d, "seq1", "frame00000.png"), rng=Fixed)
    full = mi.resize_bilinear(mi.read_png(os.path.join(d, "seq1", "frame00000.png")), 256, int(np.round(480 * 256.0 / 270)))
    assert full.shape[:2] == (256, 455)
    np.testing.assert_array_equal(img, full[31:255, 230:454])


def test_stamp_past_the_ground_truth_raises_like_the_reference(tmp_path):
    """slam/network/model_input.py:388 with :122-123: bisect_left on the ground-truth stamps may return
    len(groundtruth), and the reference then fails with IndexError; no silent reuse of the last pose."""
    d = tmp_path / "seq"
    mi.write_synthetic_sequence(str(d), n_frames=4, height=24, width=32, seed=0)
    gt_file = d / "groundtruth.txt"
    rows = [l for l in gt_file.read_text().splitlines() if l and not l.startswith("#")]
    gt_file.write_text("\n".join(rows[:-1]) + "\n")        # the last frame's stamp now lies past the ground truth
    with pytest.raises(IndexError):
        mi.get_simple_input_provider(lambda: ["seq"], base_data_dir=str(tmp_path))


def test_host_reader_arithmetic_matches_the_oracle_restatement():
    """model_input.py:437-450 (integer depth scaling, concatenation, order-1 resize): the product's host form
    (gathers) against oracle/prep_oracle.py (interpolation matrices); parity unpinned, skimage is not installed."""
    from oracle import prep_oracle
    from slam_amd.utils import synthetic as syn
    for (H, W, Ho, Wo) in ((48, 64, 22, 22), (37, 53, 16, 24), (20, 20, 20, 20), (9, 7, 30, 31)):
        rgb = syn.rgb_frames(2, H, W, seed=H, kind="noise")
        depth = syn.depth_frames(2, H, W, seed=W, kind="noise")
        depth[1] = 0
        for b in range(2):
            d = depth[b].astype(np.int64)[..., None]
            d = 255 * d // max(int(d.max()), 1)
            got = mi.resize_bilinear(np.concatenate((rgb[b], d), 2), Ho, Wo).astype(np.float32)
            np.testing.assert_allclose(got, prep_oracle.get_rgbd_file(rgb[b], depth[b], Ho, Wo), rtol=1e-6, atol=1e-4)
