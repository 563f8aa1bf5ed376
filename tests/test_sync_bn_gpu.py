"""Normalisation statistics kept over the data-parallel ranks (VGG16Model(sync_batch_norm=True)): two ranks with
half the batch each compute what one process computes on the whole batch - the reference's single-device graph
normalises over the one batch it sees (slam/network/cnn_model.py:100-113) - where per-rank statistics do not."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _spawn(args):
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SLAMHIP_DETERMINISTIC"] = "1"      # sums in a fixed order: what is left is the arithmetic, not the atomics' order
    return subprocess.Popen([sys.executable, os.path.join(HERE, "sync_bn_worker.py")] + [str(a) for a in args], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def _wait(procs):
    for p in procs:
        try:
            out, _ = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        assert p.returncode == 0, out[-3000:]


def _two_ranks(tmp_path, sync, port):
    outs = [str(tmp_path / ("r%d_s%d.npz" % (r, sync))) for r in range(2)]
    _wait([_spawn([r, 2, port, sync, outs[r]]) for r in range(2)])
    return [np.load(o) for o in outs]


def _rel(a, b):
    return float(np.linalg.norm(a.astype(np.float64) - b) / (np.linalg.norm(b.astype(np.float64)) + 1e-30))


def test_two_ranks_with_shared_statistics_match_one_process_on_the_whole_batch(tmp_path):
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    full_path = str(tmp_path / "full.npz")
    _wait([_spawn(["full", full_path])])
    full = np.load(full_path)
    port = 29600 + os.getpid() % 300
    sync = _two_ranks(tmp_path, 1, port)
    local = _two_ranks(tmp_path, 0, port + 1)
    names = [k[5:] for k in full.files if k.startswith("sums_")]
    # forward statistics: a rank holds the mean of the ranks' sums, i.e. half of the whole batch's sums.  The first
    # layer's input is the data itself; deeper layers see bfloat16 activations whose roundings a last-bit
    # difference in the statistics moves, so the agreement loosens with depth (two runs of ONE process already
    # differ that way).
    for k, name in enumerate(names):
        want = full["sums_" + name] / 2.0
        tol = 1e-5 if k == 0 else 1e-3 if k < 6 else 1e-2     # measured: 6e-8 ... 6e-6 ... 1.5e-3; per-rank statistics: 2e-2 ... 9e-2
        for r in range(2):
            assert _rel(sync[r]["sums_" + name], want) < tol, (name, r, _rel(sync[r]["sums_" + name], want))
        np.testing.assert_array_equal(sync[0]["sums_" + name], sync[1]["sums_" + name])     # one all-reduce, one value
    # without the exchange each rank normalises over its own four images: already the first layer's sums differ
    first = names[0]
    assert _rel(local[0]["sums_" + first], full["sums_" + first] / 2.0) > 10 * _rel(sync[0]["sums_" + first], full["sums_" + first] / 2.0)
    # outputs: a rank's rows are the whole batch's rows of its shard
    out_sync = np.concatenate([sync[0]["output"], sync[1]["output"]])
    out_local = np.concatenate([local[0]["output"], local[1]["output"]])
    e_sync, e_local = _rel(out_sync, full["output"]), _rel(out_local, full["output"])
    assert e_sync < 0.05 and e_sync < 0.25 * e_local, (e_sync, e_local)     # measured 1.3e-2 against 0.27
    # gradients summed over the ranks = the whole batch's gradients (the loss is a plain sum over the batch);
    # first-layer filters and batch-norm parameters see every layer's backward
    for key in ("g_conv1_1__weights", "g_conv1_1__bn", "g_out_layer__weights"):
        assert np.isfinite(sync[0][key]).all()
        np.testing.assert_array_equal(sync[0][key], sync[1][key])
        e_s, e_l = _rel(sync[0][key], full[key]), _rel(local[0][key], full[key])
        # (measured 1.1e-2 for the output layer and 0.30 for the first layer, against 0.12 and 1.0 with per-rank
        # statistics.  The sums run in a fixed order here, so the 0.30 is no atomics' order: the mean of two half
        # sums and the whole sum differ in the last bit, that moves bfloat16 roundings of the activations, and twelve
        # normalised layers over eight images amplify it.  The exchange itself is checked to 1e-2 on one layer (bfloat16
        # outputs), where nothing amplifies:
        # test_encoder_gpu.py::test_batch_norm_backward_in_two_phases_over_two_shards_matches_the_whole_batch.)
        assert e_s < 0.5 and e_s < 0.5 * e_l, (key, e_s, e_l)
