"""The training entry point end to end on synthetic TUM-shaped sequences."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_vgg16_rgbd_entry_point_runs(tmp_path, monkeypatch):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    from slam_amd.network import model_config as mc
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfg = json.load(open(os.path.join(root, "resources", "model.config")))
    cfg["train"]["slam_files"] = ["freiburg1_xyz", "freiburg1_rpy"]
    cfg["train"]["model"]["base_log_dir"] = str(tmp_path / "log")
    p = tmp_path / "model.config"
    p.write_text(json.dumps(cfg))
    monkeypatch.setenv("SLAM_MODEL_CONFIG", str(p))
    monkeypatch.setattr(mc, "_config_provider", None)
    from slam_amd.main import vgg16_rgbd
    loss = vgg16_rgbd.main(["--synthetic", str(tmp_path / "tum"), "--synthetic-frames", "4", "--epochs", "1"])
    assert np.isfinite(loss)
    # the same run with the input preparation on the GPU sees the same data
    monkeypatch.setattr(mc, "_config_provider", None)
    loss_gpu_prep = vgg16_rgbd.main(["--synthetic", str(tmp_path / "tum"), "--epochs", "1", "--gpu-prep"])
    assert np.isfinite(loss_gpu_prep)
    saved = torch.load(str(tmp_path / "log" / "checkpoints" / "learned_weights.pt"), weights_only=True)
    assert saved["conv1_1/weights"].shape == (3, 3, 4, 64) and saved["fc6-conv/weights"].shape == (25088, 4096)
    monkeypatch.setattr(mc, "_config_provider", None)
