"""Input provider of the RGB-D trainers with the reference's class, method and
attribute names (slam/network/model_input.py:322-454, 517-529).

Per sequence directory it reads `associate.txt` and `groundtruth.txt`,
precomputes the per-frame twist label from the ABSOLUTE pose (the reference never
uses its `trans_old`, model_input.py:387-394) through the C ABI's host pose
functions, and yields one frame per sequence per step: RGB + depth read, depth
scaled to `255 * d / max(d)`, 4-channel concat, bilinear resize to 224x224,
per-batch channel-mean subtraction.

Two points where the reference's libraries are not available here (parity
unpinned, behaviour chosen and stated):
  * `255 * depth / np.max(depth)` runs under Python 2 on an integer image: floor
    division.  The 16-bit PNG is widened to int32 first (no uint16 wrap).
  * `skimage.transform.resize(..., preserve_range=True)` is restated as order-1
    interpolation at half-pixel centres with edge clamping.
"""
import os
import random

import numpy as np

from .. import frame_pairs as fp
from ..preprocess.geometry import _find_label
from ..utils.time_utils import time_it


def get_mean(rgb_images, channels=3):
    """model_input.py:517-522 - mean over rows, then columns, then images."""
    total = np.zeros(channels)
    for image in rgb_images:
        total += np.mean(np.mean(image, axis=0), axis=0)
    return total / len(rgb_images)


def read_png(path):
    """8-bit RGB -> uint8 [H,W,3]; 16-bit grey -> int32 [H,W]."""
    from PIL import Image
    with Image.open(path) as im:
        a = np.array(im)
    return a.astype(np.int32) if a.ndim == 2 else a


def resize_bilinear(img, out_h, out_w):
    """[H,W,C] float -> [out_h,out_w,C]: order-1 interpolation, sample centres at
    (i + 0.5) * H / out_h - 0.5, clamped to the image."""
    img = np.asarray(img, np.float64)
    H, W = img.shape[:2]
    ys = np.clip((np.arange(out_h) + 0.5) * H / out_h - 0.5, 0, H - 1)
    xs = np.clip((np.arange(out_w) + 0.5) * W / out_w - 0.5, 0, W - 1)
    y0 = np.floor(ys).astype(int)
    x0 = np.floor(xs).astype(int)
    y1 = np.minimum(y0 + 1, H - 1)
    x1 = np.minimum(x0 + 1, W - 1)
    fy = (ys - y0)[:, None, None]
    fx = (xs - x0)[None, :, None]
    top = img[y0][:, x0] * (1 - fx) + img[y0][:, x1] * fx
    bot = img[y1][:, x0] * (1 - fx) + img[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def absolute_twist_labels(gt_rows):
    """model_input.py:384-396 - twist of the absolute pose for i > 0, zeros for i = 0
    (clamped-trace variant of _trans_to_twist, model_input.py:46-76)."""
    gt_rows = np.ascontiguousarray(gt_rows, np.float32)
    twist = np.zeros((gt_rows.shape[0], 6))
    if gt_rows.shape[0] > 1:
        T = fp.quat_to_transformation_batch(gt_rows[1:])
        twist[1:] = fp.trans_to_twist_batch(T, clamp=True)
    return twist


class QueuedInputProvider:
    """The queue-fed provider (model_input.py:126-319): per training sequence a cyclic,
    unshuffled queue of (rgb file, depth file, twist) over a window of at most 300
    frames, read through the in-graph reader's arithmetic (8-bit depth, TensorFlow
    bilinear resize, depth * 255 / max; slam_amd.network.prep.rgbd_batch_tf).

    The reference hands back graph tensors that every session.run refills; here
    get_training_batch() returns the GPU tensors once and run() refills them with the
    next element of every queue."""

    BASE_DATA_DIR = '/home/sanjeev/data/'
    FX = [517.3, 520.9, 535.4]
    FY = [516.5, 521.0, 539.2]
    CX = [318.6, 325.1, 320.1]
    CY = [255.3, 249.7, 247.6]
    FACTOR = 1

    def __init__(self, base_data_dir=None, image_reader=read_png, config_provider=None):
        from .model_config import get_config_provider
        self.config_provider = config_provider or get_config_provider()
        base = self.BASE_DATA_DIR if base_data_dir is None else base_data_dir
        self.training_filenames = [os.path.join(base, name) for name in self.config_provider.training_filenames()]
        self.batch_size = len(self.training_filenames)
        self.image_reader = image_reader
        self._queues = None

    def get_training_batch(self):
        """-> (images_batch, groundtruth_batch): per sequence one float32 GPU tensor
        [224,224,4] and one [1,1,6] (model_input.py:154-180), filled with the first element."""
        import torch
        dev = torch.device("cuda", torch.cuda.current_device())
        sequence_length = 300
        self._queues, self._images, self._twists = [], [], []
        for filename in self.training_filenames:
            associations = np.loadtxt(os.path.join(filename, "associate.txt"), dtype="str", unpack=False)
            groundtruth = np.loadtxt(os.path.join(filename, "groundtruth.txt"), dtype="str", unpack=False)
            twist, rgb_filepaths, depth_filepaths = self._get_data(associations, groundtruth, sequence_length,
                                                                   base_dir=filename)
            self._queues.append(dict(rgb=[os.path.join(filename, f) for f in rgb_filepaths],
                                     depth=[os.path.join(filename, f) for f in depth_filepaths],
                                     twist=np.asarray(twist, np.float64), pos=0))
            self._images.append(torch.zeros(224, 224, 4, dtype=torch.float32, device=dev))
            self._twists.append(torch.zeros(1, 1, 6, dtype=torch.float32, device=dev))
        self.run()
        return self._images, self._twists

    def run(self):
        """Next element of every queue into the tensors get_training_batch() returned
        (the queues wrap around, slice_input_producer without an epoch limit)."""
        for q, image, twist in zip(self._queues, self._images, self._twists):
            i = q["pos"] % len(q["rgb"])
            q["pos"] += 1
            self.read_rgbd_data((q["rgb"][i], q["depth"][i], q["twist"][i]), image, twist)

    def read_rgbd_data(self, input_queue, image=None, twist=None):
        """(rgb path, depth path, twist[6]) -> (float32 [224,224,4], float32 [1,1,6]) GPU
        tensors (model_input.py:185-217)."""
        import torch
        from . import prep
        dev = torch.device("cuda", torch.cuda.current_device())
        rgb = torch.from_numpy(np.ascontiguousarray(self.image_reader(input_queue[0]), np.uint8)[None]).to(dev)
        depth = torch.from_numpy(np.ascontiguousarray(self.image_reader(input_queue[1])).astype(np.uint16)[None]).to(dev)
        out = prep.rgbd_batch_tf(rgb, depth, (224, 224), out=None if image is None else image.view(1, 224, 224, 4))
        tw = torch.from_numpy(np.asarray(input_queue[2], np.float32).reshape(1, 1, 6)).to(dev)
        if twist is not None:
            twist.copy_(tw)
        return out[0], tw

    def get_batch_size(self):
        return self.batch_size

    def _point_cloud(self, depth_image, dataset_int):
        """model_input.py:222-229 indexes the cloud with u * v (sic), which overwrites most
        rows; the back-projection itself is the one of preprocess.geometry._point_cloud,
        which this returns (row u + v * W, compacted)."""
        from ..preprocess import geometry
        return geometry._point_cloud(depth_image, dataset_int)

    def _backtransform_pointcloud(self, pointcloud, dataset_int):
        from ..preprocess import geometry
        return geometry._backtransform_pointcloud(pointcloud, dataset_int)

    def _get_data(self, associations, groundtruth, sequence_length, dynamic_drop=0.0, static_drop=1, rng=np.random,
                  base_dir=None):
        """model_input.py:243-319: as preprocess.geometry._get_data with the clamped twist.
        The reference's drop branch cannot run (undefined names at :269, :279); here it
        runs the overlap test of preprocess.geometry._get_data.  base_dir: the sequence directory
        the depth paths of `associations` are relative to (test_import.py:204 joins them the same way)."""
        from ..preprocess import geometry
        reader = self.image_reader if base_dir is None else (lambda p: self.image_reader(os.path.join(base_dir, p)))
        return geometry._get_data(associations, groundtruth, sequence_length, dynamic_drop, static_drop,
                                  read_depth=reader, rng=rng, clamp=True)


class SimpleInputProvider:

    class SequenceBatch:

        def __init__(self):
            self.rgb_filenames = []
            self.depth_filenames = []
            self.rgbd_images = []
            self.groundtruths = []

    class SequenceBatchIterator:
        """One frame of every (sequence, offset) entry per step, offsets advancing
        (model_input.py:333-367)."""

        def __init__(self, input_provider, seqdir_vs_offset, sequence_length):
            self.counter = 0
            self.seqdir_vs_offset = seqdir_vs_offset
            self.input_provider = input_provider
            self.sequence_length = sequence_length

        def __iter__(self):
            return self

        @time_it
        def __next__(self):
            if self.counter >= self.sequence_length:
                raise StopIteration()
            batch = SimpleInputProvider.SequenceBatch()
            raw = self.input_provider.raw_frames
            for i, (seqdir, offset) in enumerate(self.seqdir_vs_offset):
                if raw:     # (uint8 rgb, uint16 depth): scaling, resize and mean are left to the GPU
                    rgb_filename, depth_filename, rgbd = self.input_provider.get_raw_rgbd_file(seqdir, offset)
                else:
                    rgb_filename, depth_filename, rgbd = self.input_provider.get_rgbd_file(seqdir, offset)
                batch.rgbd_images.append(rgbd)
                batch.groundtruths.append(self.input_provider.get_ground_truth(seqdir, offset))
                batch.rgb_filenames.append(rgb_filename)
                batch.depth_filenames.append(depth_filename)
                self.seqdir_vs_offset[i][1] = offset + 1
            if not raw:
                mean = get_mean(batch.rgbd_images, channels=4)
                batch.rgbd_images = [image - mean for image in batch.rgbd_images]
            self.counter += 1
            return batch

        next = __next__

    BASE_DATA_DIR = '/usr/data/rgbd_datasets/tum_rgbd_benchmark/'

    def __init__(self, filename_provider, base_data_dir=None, image_reader=read_png):
        base = self.BASE_DATA_DIR if base_data_dir is None else base_data_dir
        self.sequence_dirs = [os.path.join(base, name) for name in filename_provider()]
        self.image_reader = image_reader
        self.raw_frames = False     # True: batches carry raw frames for slam_amd.network.prep.rgbd_batch
        self.seq_dir_map = {}
        for seq_dir in self.sequence_dirs:
            associations = np.loadtxt(os.path.join(seq_dir, "associate.txt"), dtype="str", unpack=False)
            groundtruth = np.loadtxt(os.path.join(seq_dir, "groundtruth.txt"), dtype="str", unpack=False)
            # a frame stamp beyond the last ground-truth stamp raises IndexError, as in the reference
            # (model_input.py:388 with :122-123): bisect_left may return len(groundtruth)
            rows = np.array([groundtruth[:, 1:][_find_label(groundtruth[:, 0], stamp)]
                             for stamp in associations[:, 0]]).astype(np.float32)
            self.seq_dir_map[seq_dir] = dict(associations=associations, groundtruth=groundtruth,
                                             relpos=absolute_twist_labels(rows))

    def sequence_batch_itr(self, sequence_length, batch_size):
        """model_input.py:398-411."""
        random.shuffle(self.sequence_dirs)
        seqdir_vs_offset = []
        for i in range(batch_size):
            seq_dir = self.sequence_dirs[i % len(self.sequence_dirs)]
            total_frames = len(self.seq_dir_map[seq_dir]['associations'])
            seqdir_vs_offset.append([seq_dir, random.randint(0, total_frames - sequence_length)])
        return self.SequenceBatchIterator(self, seqdir_vs_offset, sequence_length)

    def complete_seq_iter(self):
        """All frames of one randomly chosen sequence (model_input.py:417-425)."""
        random.shuffle(self.sequence_dirs)
        seq_dir = self.sequence_dirs[0]
        total_frames = len(self.seq_dir_map[seq_dir]['associations'])
        return self.SequenceBatchIterator(self, [[seq_dir, 0]], total_frames)

    def get_rgbd_file(self, dirname, offset):
        """model_input.py:427-450 -> (rgb path, depth path, float32 [224,224,4])."""
        associations = self.seq_dir_map[dirname]['associations']
        if associations[offset, 1].startswith('depth'):
            rgb_filename = os.path.join(dirname, associations[offset, 3])
            depth_filename = os.path.join(dirname, associations[offset, 1])
        else:
            rgb_filename = os.path.join(dirname, associations[offset, 1])
            depth_filename = os.path.join(dirname, associations[offset, 3])
        rgb_img = self.image_reader(rgb_filename)
        depth_img = np.asarray(self.image_reader(depth_filename)).astype(np.int64)
        depth_img = depth_img.reshape(list(depth_img.shape) + [1])
        depth_img = 255 * depth_img // max(int(np.max(depth_img)), 1)    # Python-2 integer division
        rgbd_img = np.concatenate((rgb_img, depth_img), 2)
        rgbd_img = resize_bilinear(rgbd_img, 224, 224)
        return rgb_filename, depth_filename, rgbd_img.astype(np.float32)

    def _frame_paths(self, dirname, offset):
        associations = self.seq_dir_map[dirname]['associations']
        if associations[offset, 1].startswith('depth'):
            return os.path.join(dirname, associations[offset, 3]), os.path.join(dirname, associations[offset, 1])
        return os.path.join(dirname, associations[offset, 1]), os.path.join(dirname, associations[offset, 3])

    def get_raw_rgbd_file(self, dirname, offset):
        """(rgb path, depth path, (uint8 [H,W,3], uint16 [H,W])): the frame as stored."""
        rgb_filename, depth_filename = self._frame_paths(dirname, offset)
        rgb_img = np.ascontiguousarray(self.image_reader(rgb_filename), np.uint8)
        depth_img = np.ascontiguousarray(self.image_reader(depth_filename)).astype(np.uint16)
        return rgb_filename, depth_filename, (rgb_img, depth_img)

    def get_ground_truth(self, dirname, offset):
        return self.seq_dir_map[dirname]['relpos'][offset, :]


def read_rgb_image(filepath, image_reader=read_png, rng=np.random):
    """model_input.py:497-515 - scale so that the smaller side is 256 (output size
    round(factor * side), order-1 interpolation as resize_bilinear), then a random
    224 x 224 crop."""
    rgb_img = np.asarray(image_reader(filepath))
    width = height = 224
    factor = 256.0 / min(rgb_img.shape[0], rgb_img.shape[1])
    rgb_img = resize_bilinear(rgb_img, int(np.round(rgb_img.shape[0] * factor)), int(np.round(rgb_img.shape[1] * factor)))
    width_start = rng.randint(0, rgb_img.shape[1] - width)
    height_start = rng.randint(0, rgb_img.shape[0] - height)
    return rgb_img[height_start:height_start + height, width_start:width_start + width]


class PoseNetInputProvider:
    """Batches of the Cambridge landmarks data (model_input.py:456-495): rows
    'image x y z qw qx qy qz' of dataset_train.txt in file order, images scaled, cropped
    and mean-subtracted per batch.  As in the reference the last partial batch and one
    full batch that would end exactly at the last row are not produced (`index +
    batch_size < len`)."""

    BASE_DIR = '/usr/data/cvpr_shared/lingni/cambridge_pose_dataset/KingsCollege/'

    def __init__(self, base_dir=None, image_reader=read_png):
        self.base_dir = self.BASE_DIR if base_dir is None else base_dir
        self.image_reader = image_reader
        rows = []
        with open(os.path.join(self.base_dir, 'dataset_train.txt')) as f:
            for line in f:
                fields = line.split()
                if len(fields) == 8:        # the dataset's three title lines have other counts
                    rows.append(fields)
        self.sequence_info = np.array(rows, dtype="str").reshape(-1, 8)

    class PoseNetBatch:
        pass

    class PoseNetIterator:

        def __init__(self, provider, batch_size):
            self.provider = provider
            self.sequence_info = provider.sequence_info
            self.index = 0
            self.batch_size = batch_size

        def __iter__(self):
            return self

        def __next__(self):
            if self.index + self.batch_size >= len(self.sequence_info):
                raise StopIteration()
            batch_info = self.sequence_info[self.index:self.index + self.batch_size]
            filenames = [os.path.join(self.provider.base_dir, filename) for filename in batch_info[:, 0]]
            rgb_files = [read_rgb_image(filename, self.provider.image_reader) for filename in filenames]
            mean = get_mean(rgb_files)
            batch = PoseNetInputProvider.PoseNetBatch()
            batch.rgb_files = [rgb_file - mean for rgb_file in rgb_files]
            batch.groundtruths = batch_info[:, 1:]
            batch.rgb_filenames = filenames
            self.index += self.batch_size
            return batch

        next = __next__

    def sequence_batch_itr(self, batch_size):
        return PoseNetInputProvider.PoseNetIterator(self, batch_size)


def write_synthetic_posenet_dataset(base_dir, n_images=6, height=270, width=480, seed=0):
    """A Cambridge-landmarks-shaped directory (seq1/*.png + dataset_train.txt) of synthetic
    images and poses, for runs without the dataset."""
    from PIL import Image
    from ..utils import synthetic as syn
    os.makedirs(os.path.join(base_dir, "seq1"), exist_ok=True)
    rgb = syn.rgb_frames(n_images, height, width, seed=seed, kind="smooth")
    poses = syn.pose_pairs(n_images, seed=seed)[:, 1]       # tx ty tz qx qy qz qw
    with open(os.path.join(base_dir, "dataset_train.txt"), "w") as f:
        f.write("Visual Landmark Dataset V1\nImageFile, Camera Position [X Y Z W P Q R]\n\n")
        for i in range(n_images):
            name = "seq1/frame%05d.png" % i
            Image.fromarray(rgb[i]).save(os.path.join(base_dir, name))
            t, q = poses[i, :3], poses[i, 3:]
            f.write("%s %s\n" % (name, " ".join("%.6f" % v for v in (t[0], t[1], t[2], q[3], q[0], q[1], q[2]))))
    return base_dir


_queued_input_provider = None


def get_queued_input_provider(base_data_dir=None):
    """model_input.py:524-526 (created on first use instead of at import)."""
    global _queued_input_provider
    if _queued_input_provider is None:
        _queued_input_provider = QueuedInputProvider(base_data_dir)
    return _queued_input_provider


def get_simple_input_provider(filename_provider, base_data_dir=None):
    """model_input.py:528-529."""
    return SimpleInputProvider(filename_provider, base_data_dir=base_data_dir)


def write_synthetic_sequence(seq_dir, n_frames=12, height=480, width=640, seed=0):
    """A TUM-shaped sequence directory (rgb/*.png, depth/*.png 16-bit, associate.txt,
    groundtruth.txt) of synthetic frames, for runs without the dataset."""
    from PIL import Image
    from ..utils import synthetic as syn
    os.makedirs(os.path.join(seq_dir, "rgb"), exist_ok=True)
    os.makedirs(os.path.join(seq_dir, "depth"), exist_ok=True)
    depth = syn.depth_frames(n_frames, height, width, seed=seed, kind="plane")
    rgb = syn.rgb_frames(n_frames, height, width, seed=seed, kind="smooth")
    poses = syn.pose_pairs(n_frames, seed=seed)[:, 1]
    with open(os.path.join(seq_dir, "associate.txt"), "w") as fa, \
            open(os.path.join(seq_dir, "groundtruth.txt"), "w") as fg:
        for i in range(n_frames):
            stamp = "%.4f" % (1305031100.0 + i / 30.0)
            Image.fromarray(rgb[i]).save(os.path.join(seq_dir, "rgb", stamp + ".png"))
            Image.fromarray(depth[i]).save(os.path.join(seq_dir, "depth", stamp + ".png"))
            fa.write("%s rgb/%s.png %s depth/%s.png\n" % (stamp, stamp, stamp, stamp))
            fg.write(stamp + " " + " ".join("%.6f" % v for v in poses[i]) + "\n")
    return seq_dir
