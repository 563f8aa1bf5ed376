"""TEST INFRASTRUCTURE - CPU restatement (numpy float32) of the reference's in-graph
RGB-D reader, slam/network/model_input.py:185-217.

The arithmetic lives in TensorFlow 0.x (un-vendored, unpinned), restated from its
published definitions: tf.image.decode_png with the default uint8 dtype reduces a
16-bit sample to its upper byte; tf.image.resize_images (bilinear, no corner
alignment) samples at index * in / out with top = floor, bottom = min(ceil, in - 1)
and lerps in float32, columns first.  Parity unpinned: TensorFlow is not installed
and the reference holds no fixture for this path.
"""
import numpy as np


def tf_resize_bilinear(img, out_h, out_w):
    img = np.asarray(img, np.float32)
    H, W = img.shape[:2]
    in_y = np.arange(out_h, dtype=np.float32) * (np.float32(H) / np.float32(out_h))
    in_x = np.arange(out_w, dtype=np.float32) * (np.float32(W) / np.float32(out_w))
    y0, x0 = np.floor(in_y).astype(int), np.floor(in_x).astype(int)
    y1, x1 = np.minimum(np.ceil(in_y).astype(int), H - 1), np.minimum(np.ceil(in_x).astype(int), W - 1)
    fy = (in_y - y0.astype(np.float32))[:, None, None]
    fx = (in_x - x0.astype(np.float32))[None, :, None]
    top = img[y0][:, x0] + (img[y0][:, x1] - img[y0][:, x0]) * fx
    bot = img[y1][:, x0] + (img[y1][:, x1] - img[y1][:, x0]) * fx
    return (top + (bot - top) * fy).astype(np.float32)


def read_rgbd_data(rgb_u8, depth_u16, out_h=224, out_w=224):
    """uint8 [H,W,3], uint16 [H,W] -> float32 [out_h,out_w,4]."""
    rgb = tf_resize_bilinear(rgb_u8, out_h, out_w)
    depth = tf_resize_bilinear((np.asarray(depth_u16) >> 8)[..., None], out_h, out_w)
    with np.errstate(invalid="ignore", divide="ignore"):
        depth = depth * np.float32(255.0) / np.max(depth)
    return np.concatenate((rgb, depth), 2).astype(np.float32)
