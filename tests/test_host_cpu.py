"""CPU-only tests of the host side: the C restatement of the oracle against the
numpy oracle, the configuration provider, the encoder's FLOP model, and the
data-parallel plumbing under a 2-process gloo group.
"""
import json
import os
import socket

import numpy as np
import pytest
import torch

from oracle import slam_oracle as so

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _pairs(n, h, w, seed):
    from slam_amd.utils import synthetic as syn
    gt = syn.pose_pairs(n, seed=seed + 3)
    d1 = syn.depth_frames(n, h, w, seed=seed + 1, kind="noise")
    d2 = syn.depth_frames(n, h, w, seed=seed + 2, kind="noise")
    c1 = syn.rgb_frames(n, h, w, seed=seed + 1, kind="noise")
    c2 = syn.rgb_frames(n, h, w, seed=seed + 2, kind="noise")
    rel = np.stack([so.relative_pose(so.quat_to_transformation(gt[i, 0]), so.quat_to_transformation(gt[i, 1]))
                    for i in range(n)])
    return d1, c1, d2, c2, rel


def test_c_oracle_is_bit_identical_to_numpy_oracle():
    from oracle import c_oracle
    K = so.intrinsics(1)
    for (h, w, factor, seed) in ((24, 32, 1.0, 0), (17, 29, 5000.0, 5)):
        d1, c1, d2, c2, rel = _pairs(3, h, w, seed)
        s_np, c_np = so.pair_loss(d1, c1, d2, c2, rel, K, factor)
        s_c, c_c = c_oracle.pair_loss(d1, c1, d2, c2, rel, K, factor)
        assert np.array_equal(c_np, c_c)
        np.testing.assert_allclose(s_c, s_np, rtol=1e-12)
        P = so.warp_pair(d1[0], c1[0], d2[0], c2[0], rel[0], K, factor)
        C = c_oracle.warp_pair(d1[0], c1[0], d2[0], c2[0], rel[0], K, factor)
        assert np.array_equal(C["uvd"], P["uvd"], equal_nan=True)
        assert np.array_equal(C["iu_iv"][..., 0], P["iu"]) and np.array_equal(C["iu_iv"][..., 1], P["iv"])
        assert np.array_equal(C["warped_rgb"], P["warped_rgb"])
        assert np.array_equal(C["r_geo"], P["r_geo"])


def test_config_provider_has_the_reference_getters(tmp_path, monkeypatch):
    from slam_amd.network import model_config as mc
    cp = mc.ModelConfigProvider(os.path.join(ROOT, "resources", "model.config"))
    assert cp.training_filenames()[0] == "freiburg1_xyz" and len(cp.training_filenames()) == 10
    assert cp.test_filenames() == ["freiburg1_plant", "freiburg2_xyz", "freiburg2_rpy", "freiburg2_desk",
                                   "freiburg3_teddy"]
    assert (cp.lstm_layers(), cp.cnn_output_dim(), cp.epoch(), cp.sequence_length(), cp.batch_size()) == \
        (5, 6, 5000, 200, 1)
    assert (cp.learning_rate(), cp.normalization_epsilon(), cp.optimizer(), cp.base_log_dir()) == \
        (0.0001, 0.001, "AdamOptimizer", "")
    # an override file and the process-wide accessor
    cfg = json.load(open(os.path.join(ROOT, "resources", "model.config")))
    cfg["train"]["model"]["optimizer"] = "RMSPropOptimizer"
    p = tmp_path / "model.config"
    p.write_text(json.dumps(cfg))
    monkeypatch.setenv("SLAM_MODEL_CONFIG", str(p))
    monkeypatch.setattr(mc, "_config_provider", None)
    assert mc.get_config_provider().optimizer() == "RMSPropOptimizer"
    assert mc.get_config_provider() is mc.get_config_provider()


def test_encoder_flop_model():
    from slam_amd.network.cnn_model import model_flops
    fwd, both = model_flops(1, 224, 224)
    # 12 conv layers of the reference network (no conv3_3) + three dense layers, per image
    conv = 2 * 9 * (224 * 224 * (4 * 64 + 64 * 64) + 112 * 112 * (64 * 128 + 128 * 128) +
                    56 * 56 * (128 * 256 + 256 * 256) + 28 * 28 * (256 * 512 + 2 * 512 * 512) +
                    14 * 14 * 3 * 512 * 512)
    fc = 2 * (7 * 7 * 512 * 4096 + 4096 * 4096 + 4096 * 6)
    assert fwd == conv + fc
    assert both == 3 * fwd - 2 * 9 * 224 * 224 * 4 * 64
    assert model_flops(128, 224, 224)[1] == 128 * both


def test_shard_range_covers_everything_once():
    from slam_amd.distributed import shard_range
    for n in (0, 1, 7, 64, 513):
        for world in (1, 2, 3, 8):
            got = [shard_range(n, r, world) for r in range(world)]
            assert got[0][0] == 0 and got[-1][1] == n
            assert all(got[i][1] == got[i + 1][0] for i in range(world - 1))
            sizes = [e - b for b, e in got]
            assert max(sizes) - min(sizes) <= 1


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rank_main(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from slam_amd import distributed as dp
    try:
        # (1) frame pairs shard across ranks with no data-path collective; only the reduced
        #     totals are exchanged.  The per-pair reductions come from the oracle here (no GPU).
        n, h, w = 5, 12, 16
        d1, c1, d2, c2, rel = _pairs(n, h, w, seed=11)
        b, e = dp.shard_range(n)
        K = so.intrinsics(1)
        sums, counts = so.pair_loss(d1[b:e], c1[b:e], d2[b:e], c2[b:e], rel[b:e], K, 1.0)
        tot_s, tot_c = dp.total_pair_sums(torch.from_numpy(sums), torch.from_numpy(counts.astype(np.int64)))
        # (2) bucketed gradient mean: two asynchronous all-reduces over one flat buffer
        flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
        h1 = dp.start_bucket_allreduce(flat, 0, 4)
        h2 = dp.start_bucket_allreduce(flat, 4, 10)
        dp.finish([h1, h2])
        # (3) the trainers' per-epoch step count: every rank walks a sequence of its own (lengths differ)
        #     and all stop after the shortest, so that the per-step all-reduces keep matching
        from slam_amd.network import model_input as mi
        names = ["seq_a", "seq_b"]
        base = os.path.join(out_dir, "data%d" % rank)
        for i, nme in enumerate(names):
            mi.write_synthetic_sequence(os.path.join(base, nme), n_frames=4 + 3 * i, height=24, width=32, seed=i)
        prov = mi.get_simple_input_provider(lambda: names, base_data_dir=base)
        prov.sequence_dirs.sort()
        prov.sequence_dirs = prov.sequence_dirs[rank:] + prov.sequence_dirs[:rank]   # rank r starts on sequence r
        import random
        random.shuffle = lambda x: None      # complete_seq_iter() then takes sequence_dirs[0]
        it = prov.complete_seq_iter()
        own = it.sequence_length
        it.sequence_length = dp.agree_min(it.sequence_length)
        steps = 0
        for _ in it:
            g = torch.ones(3) * (rank + 1)
            dist.all_reduce(g)               # stands for the step's gradient all-reduce
            steps += 1
        # (4) the gradient buckets of the training step: ranges become final in the order of the flat buffer,
        #     a bucket starts once 2 Mi elements are ready (or on request), the rest goes last
        n_grad = (5 << 20) + 77
        grad = torch.arange(n_grad, dtype=torch.float32) % 1000 * (rank + 1)
        buckets = dp.BucketedAllReduce(grad)
        buckets.ready(1 << 20, force=True)            # "fully connected layers done"
        for end in ((1 << 20) + (3 << 20), (1 << 20) + (3 << 20) + 1000, n_grad - 5):
            buckets.ready(end)                          # "a block of convolution layers done"
        started = buckets.finish(n_grad)
        dp.require_equal_over_ranks(n_grad, "gradient elements")
        unequal = ""
        try:
            dp.require_equal_over_ranks(4 + rank, "batch size per rank")
        except RuntimeError as ex:
            unequal = str(ex)
        np.savez(os.path.join(out_dir, "rank%d.npz" % rank), tot_s=tot_s.numpy(), tot_c=tot_c.numpy(),
                 flat=flat.numpy(), shard=np.array([b, e]), own_len=own, steps=steps, grad=grad.numpy(),
                 started=np.array(started), unequal=np.array(unequal))
    finally:
        dist.destroy_process_group()


def test_two_rank_gloo_sharding_and_reductions(tmp_path):
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    mp.spawn(_rank_main, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    n, h, w = 5, 12, 16
    d1, c1, d2, c2, rel = _pairs(n, h, w, seed=11)
    sums, counts = so.pair_loss(d1, c1, d2, c2, rel, so.intrinsics(1), 1.0)
    want_s, want_c = sums.reshape(-1, 4).sum(0), counts.reshape(-1, 4).astype(np.int64).sum(0)
    shards = []
    for r in range(world):
        g = np.load(os.path.join(str(tmp_path), "rank%d.npz" % r))
        np.testing.assert_allclose(g["tot_s"], want_s, rtol=1e-12)
        assert np.array_equal(g["tot_c"], want_c)
        np.testing.assert_array_equal(g["flat"], np.arange(10, dtype=np.float32) * 3)   # (1 + 2) x
        shards.append(tuple(g["shard"]))
        assert int(g["steps"]) == 4 and int(g["own_len"]) == (4, 7)[r]    # both ranks stop after the shorter sequence
        n_grad = (5 << 20) + 77
        np.testing.assert_array_equal(g["grad"], (np.arange(n_grad) % 1000 * 3).astype(np.float32))   # (1 + 2) x, every element once
        assert [tuple(x) for x in g["started"]] == [(0, 1 << 20), (1 << 20, 4 << 20), (4 << 20, n_grad)]
        assert "batch size per rank differs between the ranks (4 .. 5" in str(g["unequal"])
    assert shards == [(0, 3), (3, 5)]
