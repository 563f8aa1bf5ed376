"""GoogLeNet (PoseNet) trainers: the reference's slam/main/posenet.py:13-57,
posenet_noLRN.py and posenet_rgbd.py:15-61 on the MI355X kernels.

  python -m slam_amd.main.posenet --synthetic /tmp/kings --epochs 1 --batch-size 2
  python -m slam_amd.main.posenet_noLRN ...        (no LRN layers, batch size 1)
  python -m slam_amd.main.posenet_rgbd --synthetic /tmp/tum ...   (4-channel RGB-D input, 6-wide twist)

Same flow as the reference: the three-head network, `add_l2_loss`, Adam on
1e-5 * 0.1^(step // 10), weights saved every 10 epochs.
"""
import argparse
import logging
import os

import numpy as np
import torch

from slam_amd.network import checkpoint
from slam_amd.network.model_config import get_config_provider
from slam_amd.network.model_input import (PoseNetInputProvider, get_simple_input_provider,
                                          write_synthetic_posenet_dataset, write_synthetic_sequence)


def run(argv, google_net_class, rgbd, default_batch_size, default_base_dir):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--synthetic", default=None, metavar="DIR", help="write a synthetic dataset into DIR and train on it")
    ap.add_argument("--synthetic-items", type=int, default=6, help="images (PoseNet data) or frames per sequence (RGB-D)")
    ap.add_argument("--base-dir", default=default_base_dir, help="where checkpoints/ is written")
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--batch-size", type=int, default=default_batch_size)
    ap.add_argument("--sequence-length", type=int, default=200)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    logger = logging.getLogger("slam_amd")
    if not torch.cuda.is_available():
        raise RuntimeError("the GoogLeNet trainers need a GPU (no CPU fallback)")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    img_h = img_w = 224
    batch_size = a.batch_size
    data_dir = a.data_dir
    if rgbd:
        config_provider = get_config_provider()
        if a.synthetic:
            data_dir = a.synthetic
            for i, name in enumerate(config_provider.training_filenames()):
                if not os.path.exists(os.path.join(data_dir, name, "associate.txt")):
                    write_synthetic_sequence(os.path.join(data_dir, name), a.synthetic_items, seed=i)
        input_provider = get_simple_input_provider(config_provider.training_filenames, base_data_dir=data_dir)
        channels, out_dim = 4, 6
        batches = lambda: input_provider.sequence_batch_itr(a.sequence_length, batch_size)
        images = lambda batch: batch.rgbd_images
    else:
        if a.synthetic:
            data_dir = a.synthetic
            if not os.path.exists(os.path.join(data_dir, "dataset_train.txt")):
                write_synthetic_posenet_dataset(data_dir, a.synthetic_items)
        input_provider = PoseNetInputProvider(data_dir)
        channels, out_dim = 3, 7
        batches = lambda: input_provider.sequence_batch_itr(batch_size)
        images = lambda batch: batch.rgb_files
    weights_file = os.path.join(a.base_dir, 'checkpoints', 'learned_weights.pt')

    rgb_input_batch = torch.zeros(batch_size, img_h, img_w, channels, dtype=torch.float32, device=dev)
    groundtruth_batch = torch.zeros(batch_size, out_dim, dtype=torch.float32, device=dev)
    google_net = google_net_class({'data': rgb_input_batch}, out_dim)
    loss = google_net.add_l2_loss(groundtruth_batch)
    apply_gradient_op = google_net.add_optimizer()

    loss_value = float("nan")
    for step in range(a.epochs):
        logger.info('Executing step:%d', step)
        for i, sequence_batch in enumerate(batches()):
            rgb_input_batch.copy_(torch.from_numpy(np.stack(images(sequence_batch)).astype(np.float32)))
            groundtruth_batch.copy_(torch.from_numpy(np.asarray(sequence_batch.groundtruths).astype(np.float32)))
            apply_gradient_op()
            loss_value = float(loss)
            logger.info('epoc:%d, seq_no:%d loss :%s', step, i, loss_value)
        if step % 10 == 0:
            logger.info('Saving weights.')
            checkpoint.save(weights_file, google_net.named_variables(), google_net.global_step)
        logger.info('epoc:%d, loss:%s', step, loss_value)
    return loss_value


def main(argv=None):
    from slam_amd.network.google_net import GoogleNet
    return run(argv, GoogleNet, rgbd=False, default_batch_size=20, default_base_dir='/usr/prakt/s085/posenet/')


if __name__ == '__main__':
    main()
