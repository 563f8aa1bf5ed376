"""The functions of the reference's early data-import script
(slam/preprocess/import_data.py:8-57) with their names; the script's module-level
TensorFlow session loop is not reproduced.

`_absolute_position` runs in the library's host pose code.  `_image_preprocessing`
keeps the script's arithmetic (depth shifted to start at zero and scaled to 0..255,
4-channel concat, scipy.ndimage.zoom to 224 x 224) on numpy arrays instead of
evaluated TensorFlow tensors."""
import bisect

import numpy as np

from .. import frame_pairs as fp


def _find_label(groundtruth, timestamp):
    """import_data.py:8-9"""
    return bisect.bisect_left(groundtruth, timestamp)


def _absolute_position(groundtruth):
    """import_data.py:12-42: [x, y, z, rotation vector] of one ground-truth row."""
    return fp.absolute_position_batch(np.asarray(groundtruth, dtype=np.float64)[:7])[0]


def _image_preprocessing(image_rgb, image_depth):
    """import_data.py:45-57: uint8 [480,640,3] + depth [480,640,1] -> float [224,224,4]."""
    import scipy.ndimage
    img_rgb = np.asarray(image_rgb)
    img_depth = np.array(image_depth, dtype=np.float64)
    img_depth -= img_depth.min()
    img_depth = img_depth * 255.0 / img_depth.max()
    img = np.concatenate((img_rgb, img_depth), axis=2)
    return scipy.ndimage.zoom(img, (224.0 / img.shape[0], 224.0 / img.shape[1], 1))
