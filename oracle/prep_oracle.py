"""TEST INFRASTRUCTURE - CPU restatements of the reference's two RGB-D readers: the in-graph one
(numpy float32), slam/network/model_input.py:185-217, and the host one of SimpleInputProvider.get_rgbd_file
(float64), slam/network/model_input.py:437-450 (its batch mean, :517-522, is slam_oracle.get_mean).

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


# ---- SimpleInputProvider.get_rgbd_file (model_input.py:437-450) ----
# The resize lives in skimage.transform.resize (un-vendored, unpinned; the reference passes preserve_range=True
# and leaves the order at its default 1).  Restated as order-1 interpolation at half-pixel sample centres with
# edge clamping and no smoothing, written here as two interpolation matrices (rows, then columns) rather than
# as the gathers of the product's host and GPU forms.  Parity unpinned: skimage is not installed and the
# reference holds no fixture for this path.

def _order1_matrix(n_in, n_out):
    """[n_out, n_in]: row i holds the two weights of sample centre (i + 0.5) * n_in / n_out - 0.5, clamped."""
    M = np.zeros((n_out, n_in))
    for i in range(n_out):
        c = min(max((i + 0.5) * n_in / n_out - 0.5, 0.0), n_in - 1.0)
        lo = int(np.floor(c))
        hi = min(lo + 1, n_in - 1)
        M[i, lo] += 1.0 - (c - lo)
        M[i, hi] += c - lo
    return M


def resize_order1(img, out_h, out_w):
    img = np.asarray(img, np.float64)
    My, Mx = _order1_matrix(img.shape[0], out_h), _order1_matrix(img.shape[1], out_w)
    rows = np.tensordot(My, img, axes=(1, 0))                          # [out_h, W, C]
    return np.tensordot(Mx, rows, axes=(1, 1)).transpose(1, 0, 2)      # [out_h, out_w, C]


def get_rgbd_file(rgb_u8, depth_u16, out_h=224, out_w=224):
    """uint8 [H,W,3], uint16 [H,W] -> float32 [out_h,out_w,4]: depth 255 * d / max(d) in integers (Python 2's
    `/` on integer arrays; a frame without depth keeps zeros), concatenation, resize, cast."""
    d = np.asarray(depth_u16).astype(np.int64)[..., None]
    d = 255 * d // max(int(d.max()), 1)
    return resize_order1(np.concatenate((np.asarray(rgb_u8).astype(np.int64), d), 2), out_h, out_w).astype(np.float32)

