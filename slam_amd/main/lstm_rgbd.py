"""CNN-LSTM with the RGB-D dataset: the second training entry point of the
reference (slam/main/lstm_rgbd.py:16-118) on the MI355X kernels.

  python -m slam_amd.main.lstm_rgbd --synthetic /tmp/tum --epochs 1 --sequence-length 4
  python -m slam_amd.main.lstm_rgbd --evaluate --synthetic /tmp/tum

start_training(): per epoch one `sequence_batch_itr(sequence_length, batch_size)` pass,
loss weight zero for the first frame of a sequence and identity afterwards, gradient
descent on encoder and head together, weights saved after every epoch.
evaluate_model(): restores the weights and accumulates the squared error of the
predicted twist over ten passes of 400 frames at batch size 1; the reference's call
there omits normalization_epsilon (lstm_rgbd.py:94, a TypeError as written) - it is
passed here.
"""
import argparse
import logging
import os

import numpy as np
import torch

from slam_amd.network import checkpoint
from slam_amd.network.cnn_model import VGG16Model
from slam_amd.network.lstm_model import LSTMmodel
from slam_amd.network.model_config import get_config_provider
from slam_amd.network.model_input import get_simple_input_provider, write_synthetic_sequence

logger = logging.getLogger("slam_amd")
LEARNED_WEIGHTS_FILENAME = 'resources/learned_weights.pt'
img_h = 224
img_w = 224


def _device():
    if not torch.cuda.is_available():
        raise RuntimeError("slam_amd.main.lstm_rgbd needs a GPU (no CPU fallback)")
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    return torch.device("cuda", local)


def build_complete_network(rgbd_input_batch, groundtruth_batch, lstm_init_state, batch_size, lstm_layers, cnn_output_dim,
                           normalization_epsilon, **encoder_options):
    """lstm_rgbd.py:71-77 -> (cnn_model, cnn_output, lstm_model, lstm_output)."""
    cnn_model = VGG16Model(batch_size, rgbd_input_batch, cnn_output_dim, normalization_epsilon, **encoder_options)
    cnn_output = cnn_model.build_graph()
    lstm_model = LSTMmodel(cnn_output, layer_size=cnn_output_dim, layers=lstm_layers, output_dim=6,
                           ground_truth=groundtruth_batch, batch_size=batch_size, init_state=lstm_init_state,
                           upstream=cnn_model)
    lstm_output = lstm_model.build_graph()
    return cnn_model, cnn_output, lstm_model, lstm_output


def _fill(rgbd_input_batch, groundtruth_batch, sequence_batch):
    rgbd_input_batch.copy_(torch.from_numpy(np.stack(sequence_batch.rgbd_images).astype(np.float32)))
    groundtruth_batch.copy_(torch.from_numpy(np.stack(sequence_batch.groundtruths).astype(np.float32)))


def start_training(data_dir=None, epochs=None, sequence_length=None, weights_file=LEARNED_WEIGHTS_FILENAME,
                   **encoder_options):
    dev = _device()
    config_provider = get_config_provider()
    epoch = epochs or config_provider.epoch()
    batch_size = config_provider.batch_size()
    sequence_length = sequence_length or config_provider.sequence_length()
    lstm_layers = config_provider.lstm_layers()
    cnn_output_dim = config_provider.cnn_output_dim()
    normalization_epsilon = config_provider.normalization_epsilon()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    rgbd_input_batch = torch.zeros(batch_size, img_h, img_w, 4, dtype=torch.float32, device=dev)
    groundtruth_batch = torch.zeros(batch_size, 6, dtype=torch.float32, device=dev)
    lstm_init_state = torch.zeros(batch_size, 2 * cnn_output_dim * lstm_layers, dtype=torch.float32, device=dev)
    network = build_complete_network(rgbd_input_batch, groundtruth_batch, lstm_init_state, batch_size, lstm_layers,
                                     cnn_output_dim, normalization_epsilon, **encoder_options)
    lstm_model = network[2]
    loss_weight = torch.zeros(6, 6, dtype=torch.float32, device=dev)
    loss = lstm_model.add_loss(loss_weight)
    apply_gradient_op = lstm_model.add_optimizer()

    input_provider = get_simple_input_provider(config_provider.training_filenames, base_data_dir=data_dir)
    loss_value = float("nan")
    for step in range(epoch):
        logger.info('Executing epoc:%d', step)
        for i, sequence_batch in enumerate(input_provider.sequence_batch_itr(sequence_length, batch_size)):
            _fill(rgbd_input_batch, groundtruth_batch, sequence_batch)
            loss_weight.copy_(torch.zeros(6, 6) if i == 0 else torch.eye(6))
            apply_gradient_op()
            loss_value = float(loss)
            logger.info('epoc:%d, sequence number:%d, loss:%s', step, i, loss_value)
        if int(os.environ.get("RANK", "0")) == 0:
            checkpoint.save(weights_file, lstm_model.named_variables(), lstm_model.global_step)
        logger.info('epoc:%d, loss:%s', step, loss_value)
    if world > 1:
        dist.destroy_process_group()
    return loss_value


def evaluate_model(data_dir=None, passes=10, sequence_length=400, weights_file=LEARNED_WEIGHTS_FILENAME,
                   **encoder_options):
    dev = _device()
    config_provider = get_config_provider()
    input_provider = get_simple_input_provider(config_provider.test_filenames, base_data_dir=data_dir)
    batch_size = 1
    lstm_layers = config_provider.lstm_layers()
    cnn_output_dim = config_provider.cnn_output_dim()

    rgbd_input_batch = torch.zeros(batch_size, img_h, img_w, 4, dtype=torch.float32, device=dev)
    groundtruth_batch = torch.zeros(batch_size, 6, dtype=torch.float32, device=dev)
    lstm_init_state = torch.zeros(batch_size, 2 * cnn_output_dim * lstm_layers, dtype=torch.float32, device=dev)
    network = build_complete_network(rgbd_input_batch, groundtruth_batch, lstm_init_state, batch_size, lstm_layers,
                                     cnn_output_dim, config_provider.normalization_epsilon(), **encoder_options)
    lstm_model, lstm_output = network[2], network[3]
    checkpoint.restore(weights_file, lstm_model.named_variables(), models=(network[0],))

    total_rmse = 0
    for step in range(passes):
        logger.info('Executing evaluation step:%d ', step)
        for sequence_batch in input_provider.sequence_batch_itr(sequence_length, batch_size):
            _fill(rgbd_input_batch, groundtruth_batch, sequence_batch)
            lstm_model.forward()
            result = lstm_output.cpu().numpy().astype(np.float64)
            rmse = (result - np.stack(sequence_batch.groundtruths)) ** 2
            total_rmse = total_rmse + rmse
            logger.info('Input frame info: rgb file:%s, depth file:%s, groundtruth:%s, predicted params:%s, rmse:%s ',
                        sequence_batch.rgb_filenames, sequence_batch.depth_filenames, sequence_batch.groundtruths,
                        result, rmse)
    logger.info('Total rmse on test data:%s', total_rmse)
    return total_rmse


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--synthetic", default=None, metavar="DIR",
                    help="write synthetic TUM-shaped sequences for the configured names into DIR and use them")
    ap.add_argument("--synthetic-frames", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=0)
    ap.add_argument("--sequence-length", type=int, default=0)
    ap.add_argument("--weights", default=LEARNED_WEIGHTS_FILENAME)
    ap.add_argument("--evaluate", action="store_true", help="run evaluate_model() instead of start_training()")
    ap.add_argument("--passes", type=int, default=10)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    data_dir = a.data_dir
    if a.synthetic:
        data_dir = a.synthetic
        cp = get_config_provider()
        names = cp.test_filenames() if a.evaluate else cp.training_filenames()
        marker = os.path.join(data_dir, ".written")
        if int(os.environ.get("LOCAL_RANK", "0")) == 0:
            for i, name in enumerate(names):
                if not os.path.exists(os.path.join(data_dir, name, "associate.txt")):
                    write_synthetic_sequence(os.path.join(data_dir, name), a.synthetic_frames, seed=i)
            open(marker, "w").close()
        else:                      # the other ranks of a data-parallel run wait for rank 0's files
            import time
            deadline = time.time() + 300
            while not os.path.exists(marker):
                if time.time() > deadline:
                    raise RuntimeError("synthetic sequences were not written within 300 s")
                time.sleep(0.5)
    if a.evaluate:
        return evaluate_model(data_dir, passes=a.passes, sequence_length=a.sequence_length or 400, weights_file=a.weights)
    return start_training(data_dir, epochs=a.epochs or None, sequence_length=a.sequence_length or None,
                          weights_file=a.weights)


if __name__ == '__main__':
    main()
