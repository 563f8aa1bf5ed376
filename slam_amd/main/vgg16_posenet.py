"""VGG16 on the PoseNet (Cambridge landmarks) data: the reference's
slam/main/vgg16_posenet.py:16-92 on the MI355X kernels.

  python -m slam_amd.main.vgg16_posenet --synthetic /tmp/kings --epochs 1

A [batch, 224, 224, 3] colour input, a 7-wide output [x y z qw qx qy qz], the loss
sum((x^ - x)^2) + 1100 sum((q^ - q)^2), Adam on 0.01 * 0.1^(step // 5), weights
saved every 10 epochs.
"""
import argparse
import logging
import os

import numpy as np
import torch

from slam_amd.network import checkpoint
from slam_amd.network import encoder_ops as ops
from slam_amd.network.cnn_model import VGG16Model
from slam_amd.network.model_input import PoseNetInputProvider, write_synthetic_posenet_dataset


def add_posenet_loss(vgg_model, groundtruth, beta=1100.0):
    """vgg16_posenet.py:16-25 on the encoder's output; returns the [1] loss tensor."""
    return vgg_model.add_custom_loss(
        lambda output, loss, d_out: ops.posenet_loss(output, groundtruth, loss, d_out, beta=beta))


def add_optimizer(vgg_model):
    """vgg16_posenet.py:27-41 - Adam whatever the configuration names."""
    return vgg_model.add_optimizer('AdamOptimizer')


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", default=None, help="directory holding dataset_train.txt (default: the reference's)")
    ap.add_argument("--synthetic", default=None, metavar="DIR", help="write a synthetic dataset into DIR and train on it")
    ap.add_argument("--synthetic-images", type=int, default=6)
    ap.add_argument("--base-dir", default='/home/sanjeev/posenet/', help="where checkpoints/ is written")
    ap.add_argument("--epochs", type=int, default=1000)
    ap.add_argument("--batch-size", type=int, default=1)
    a = ap.parse_args(argv)
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    logger = logging.getLogger("slam_amd")
    if not torch.cuda.is_available():
        raise RuntimeError("slam_amd.main.vgg16_posenet needs a GPU (no CPU fallback)")
    dev = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev)
    img_h = img_w = 224
    data_dir = a.data_dir
    if a.synthetic:
        data_dir = a.synthetic
        if not os.path.exists(os.path.join(data_dir, "dataset_train.txt")):
            write_synthetic_posenet_dataset(data_dir, a.synthetic_images)
    input_provider = PoseNetInputProvider(data_dir)
    weights_file = os.path.join(a.base_dir, 'checkpoints', 'learned_weights.pt')
    batch_size = a.batch_size
    normalization_epsilon = 0.001

    rgbd_input_batch = torch.zeros(batch_size, img_h, img_w, 3, dtype=torch.float32, device=dev)
    groundtruth_batch = torch.zeros(batch_size, 7, dtype=torch.float32, device=dev)
    vgg_model = VGG16Model(batch_size, rgbd_input_batch, 7, normalization_epsilon)
    vgg_model.build_graph()
    loss = add_posenet_loss(vgg_model, groundtruth_batch)
    apply_gradient_op = add_optimizer(vgg_model)

    loss_value = float("nan")
    for step in range(a.epochs):
        logger.info('Executing step:%d', step)
        for i, sequence_batch in enumerate(input_provider.sequence_batch_itr(batch_size)):
            rgbd_input_batch.copy_(torch.from_numpy(np.stack(sequence_batch.rgb_files).astype(np.float32)))
            groundtruth_batch.copy_(torch.from_numpy(sequence_batch.groundtruths.astype(np.float32)))
            apply_gradient_op()
            loss_value = float(loss)
            logger.info('epoc:%d, seq_no:%d loss :%s', step, i, loss_value)
        if step % 10 == 0:
            logger.info('Saving weights.')
            checkpoint.save(weights_file, vgg_model.params, vgg_model.global_step)
        logger.info('epoc:%d, loss:%s', step, loss_value)
    return loss_value


if __name__ == '__main__':
    main()
