"""VGG16 with the RGB-D dataset: the training entry point of the reference
(slam/main/vgg16_rgbd.py:13-66) on the MI355X kernels.

  python -m slam_amd.main.vgg16_rgbd                       # resources/model.config, TUM directories
  python -m slam_amd.main.vgg16_rgbd --synthetic /tmp/tum --max-steps 8
  python -m torch.distributed.run --nproc-per-node 8 -m slam_amd.main.vgg16_rgbd ...   # data parallel

Same flow as the reference: configuration from resources/model.config, a
[batch, 224, 224, 4] input and a [batch, 6] ground-truth buffer refilled per step
(the reference's placeholders), loss weight zero for the first batch of a sequence
and identity afterwards, weights saved every 10 epochs.  TensorFlow summaries and
the checkpoint format are not reproduced (slam_amd/network/checkpoint.py).
"""
import argparse
import logging
import os

import numpy as np
import torch

from slam_amd import distributed as dp
from slam_amd.network import checkpoint
from slam_amd.network.cnn_model import VGG16Model
from slam_amd.network.model_config import get_config_provider
from slam_amd.network import prep
from slam_amd.network.model_input import get_simple_input_provider, write_synthetic_sequence


def main(argv=None, model_class=VGG16Model):
    ap = argparse.ArgumentParser()
    ap.add_argument("--data-dir", default=None, help="directory holding the sequence directories "
                                                     "(default: the reference's BASE_DATA_DIR)")
    ap.add_argument("--synthetic", default=None, metavar="DIR",
                    help="write synthetic TUM-shaped sequences for the configured names into DIR and train on them")
    ap.add_argument("--synthetic-frames", type=int, default=8)
    ap.add_argument("--gpu-prep", action="store_true",
                    help="upload raw frames and do depth scaling, resize and mean subtraction on the GPU")
    ap.add_argument("--max-steps", type=int, default=0, help="stop after this many training steps (0 = no limit)")
    ap.add_argument("--epochs", type=int, default=0, help="override the configured number of epochs")
    ap.add_argument("--sync-batch-norm", action="store_true",
                    help="under torch.distributed: normalise over the whole job's batch (statistics averaged over the "
                         "ranks) instead of each rank's own")
    a = ap.parse_args(argv)

    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    logger = logging.getLogger("slam_amd")
    img_h = img_w = 224
    config_provider = get_config_provider()
    base_dir = config_provider.base_log_dir()
    weights_file = os.path.join(base_dir, 'checkpoints', 'learned_weights.pt')
    epoch = a.epochs or config_provider.epoch()
    batch_size = config_provider.batch_size()
    normalization_epsilon = config_provider.normalization_epsilon()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    dev = torch.device("cuda", local)

    data_dir = a.data_dir
    if a.synthetic:
        data_dir = a.synthetic
        if local == 0:
            for i, name in enumerate(config_provider.training_filenames()):
                if not os.path.exists(os.path.join(data_dir, name, "associate.txt")):
                    write_synthetic_sequence(os.path.join(data_dir, name), a.synthetic_frames, seed=i)
        if world > 1:
            dist.barrier()

    rgbd_input_batch = torch.zeros(batch_size, img_h, img_w, 4, dtype=torch.float32, device=dev)
    groundtruth_batch = torch.zeros(batch_size, 6, dtype=torch.float32, device=dev)
    loss_weight = torch.zeros(6, 6, dtype=torch.float32, device=dev)

    vgg_model = model_class(batch_size, rgbd_input_batch, 6, normalization_epsilon)
    vgg_model.sync_batch_norm = bool(a.sync_batch_norm)
    vgg_model.build_graph()
    loss = vgg_model.add_loss(loss_weight, groundtruth_batch)
    apply_gradient_op = vgg_model.add_optimizer()

    input_provider = get_simple_input_provider(config_provider.training_filenames, base_data_dir=data_dir)
    input_provider.raw_frames = a.gpu_prep
    steps, loss_value = 0, float("nan")
    for step in range(epoch):
        logger.info('Executing step:%d', step)
        sequence_iter = input_provider.complete_seq_iter()
        if world > 1:
            # every rank walks a sequence of its own choice (its data shard), and the sequences differ in
            # length: all ranks stop after the shortest, or the gradient all-reduces would stop matching
            sequence_iter.sequence_length = dp.agree_min(sequence_iter.sequence_length, device=dev)
        for i, sequence_batch in enumerate(sequence_iter):
            if a.gpu_prep:
                rgb = torch.from_numpy(np.stack([f[0] for f in sequence_batch.rgbd_images])).to(dev)
                depth = torch.from_numpy(np.stack([f[1] for f in sequence_batch.rgbd_images])).to(dev)
                prep.rgbd_batch(rgb, depth, (img_h, img_w), out=rgbd_input_batch)
            else:
                rgbd_input_batch.copy_(torch.from_numpy(np.stack(sequence_batch.rgbd_images)))
            groundtruth_batch.copy_(torch.from_numpy(np.stack(sequence_batch.groundtruths).astype(np.float32)))
            # the first frame of a sequence has no motion label: zero weight (vgg16_rgbd.py:51)
            loss_weight.copy_(torch.zeros(6, 6) if i == 0 else torch.eye(6))
            apply_gradient_op()
            loss_value = float(loss)
            logger.info('epoc:%d, seq_no:%d loss :%s', step, i, loss_value)
            steps += 1
            if a.max_steps and steps >= a.max_steps:
                break
        if step % 10 == 0 and local == 0:
            logger.info('Saving weights.')
            checkpoint.save(weights_file, dict(vgg_model.params, **checkpoint.optimizer_slots(vgg_model)),
                            vgg_model.global_step)
        logger.info('epoc:%d, loss:%s', step, loss_value)
        if a.max_steps and steps >= a.max_steps:
            break
    if world > 1:
        dist.destroy_process_group()
    return loss_value


if __name__ == '__main__':
    main()
