"""Configuration provider with the reference's getters
(slam/network/model_config.py:4-48): one JSON file, sections train/test,
read once per process.

The file is looked up as `resources/model.config` relative to the working
directory (what the reference does), then next to this package's repository
root; SLAM_MODEL_CONFIG overrides both.
"""
import json
import os

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _config_path():
    env = os.environ.get("SLAM_MODEL_CONFIG")
    if env:
        return env
    local = os.path.join("resources", "model.config")
    if os.path.exists(local):
        return local
    return os.path.join(_REPO_ROOT, "resources", "model.config")


class ModelConfigProvider:

    def __init__(self, path=None):
        with open(path or _config_path(), encoding="utf-8") as f:
            self.config = json.load(f)

    def training_filenames(self):
        return self.config['train']['slam_files']

    def test_filenames(self):
        return self.config['test']['slam_files']

    def _model(self, key):
        return self.config['train']['model'][key]

    def lstm_layers(self):
        return self._model('lstm_layers')

    def cnn_output_dim(self):
        return self._model('cnn_output_dim')

    def epoch(self):
        return self._model('epoch')

    def sequence_length(self):
        return self._model('sequence_length')

    def batch_size(self):
        return self._model('batch_size')

    def learning_rate(self):
        return self._model('learning_rate')

    def normalization_epsilon(self):
        return self._model('normalization_epsilon')

    def optimizer(self):
        return self._model('optimizer')

    def base_log_dir(self):
        return self._model('base_log_dir')


_config_provider = None


def get_config_provider():
    """The process-wide provider (model_config.py:45-48), created on first use."""
    global _config_provider
    if _config_provider is None:
        _config_provider = ModelConfigProvider()
    return _config_provider
