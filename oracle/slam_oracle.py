"""CPU oracle (numpy, float64) for the RGB-D frame-pair hot path.

TEST INFRASTRUCTURE ONLY.  This module restates, bug for bug, the numpy
arithmetic of the reference's frame-pair geometry and label maths, and defines
the dense extension (pixel indices, bilinear warp, residual sums) that the
reference does not have.  It may be imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg - as the checker -
and by nothing that ships: the product path is the HIP library and must fail
loudly when that library is missing.

Parity status: every function marked [pinned] below is checked against golden
vectors produced by running the reference's own functions (tests/golden/*.npz,
generator oracle/make_golden.py).  Functions marked [extension] have no
counterpart in the reference; they are derived from the pinned ones so that the
pinned intermediates (xyz, transformed xyz, uvd, in-image predicate, overlap)
fall out unchanged, and their parity is "unpinned by the reference".

Reference files followed (paths relative to the reference tree):
  slam/preprocess/test_import.py:14-110,148-246,255-263   (authoritative geometry)
  slam/network/model_input.py:46-76,113-119,384-396,517-522 (label variants, mean)
  slam/network/cnn_model.py:139-144, slam/network/lstm_model.py:52-55,
  slam/main/vgg16_posenet.py:16-25                        (losses)
"""
import bisect

import numpy as np

# slam/preprocess/test_import.py:255-263 - TUM freiburg1/2/3 pinhole intrinsics;
# FACTOR is 1 there (depth stays in raw sensor units; 5000 is commented out).
FX = (517.3, 520.9, 535.4)
FY = (516.5, 521.0, 539.2)
CX = (318.6, 325.1, 320.1)
CY = (255.3, 249.7, 247.6)
FACTOR = 1.0

INT32_MIN = -2147483648
INT32_MAX = 2147483647


def intrinsics(dataset_int):
    """(fx, fy, cx, cy) of TUM dataset family 1..3 (test_import.py:255-261)."""
    i = dataset_int - 1
    return (FX[i], FY[i], CX[i], CY[i])


# --------------------------------------------------------------------------
# exact fused multiply-add on float64 arrays
# --------------------------------------------------------------------------
def _two_sum(a, b):
    s = a + b
    bb = s - a
    return s, (a - (s - bb)) + (b - bb)


def _two_prod(a, b):
    p = a * b
    c = 134217729.0  # 2**27 + 1 (Veltkamp split)
    ta = c * a
    ah = ta - (ta - a)
    al = a - ah
    tb = c * b
    bh = tb - (tb - b)
    bl = b - bh
    return p, ((ah * bh - p) + ah * bl + al * bh) + al * bl


def fma(a, b, c):
    """Correctly rounded a*b+c for float64 arrays (no overflow/underflow in the
    intermediates), by error-free transformations and one round-to-odd sum
    (Boldo & Melquiond, "Emulation of FMA and correctly rounded sums", 2008).

    Needed because the reference's `np.dot(pointcloud, R)` is a BLAS dgemm whose
    x86 micro-kernels accumulate over k with hardware FMA: the observable result
    is fma(z, R2j, fma(y, R1j, x*R0j)).  Non-finite inputs fall back to a*b+c.
    """
    a, b, c = np.broadcast_arrays(np.asarray(a, np.float64), np.asarray(b, np.float64),
                                  np.asarray(c, np.float64))
    with np.errstate(all="ignore"):
        plain = a * b + c
        uh, ul = _two_prod(a, b)
        th, tl = _two_sum(c, ul)
        vh, vl = _two_sum(uh, th)
        # z = round-to-odd(tl + vl)
        z, e = _two_sum(tl, vl)
        even = (z.view(np.int64) & 1) == 0
        fix = (e != 0) & even
        z = np.where(fix, np.nextafter(z, np.where(e > 0, np.inf, -np.inf)), z)
        r = vh + z
    ok = np.isfinite(plain) & np.isfinite(r)
    return np.where(ok, r, plain)


# --------------------------------------------------------------------------
# labels / pose maths                                                  [pinned]
# --------------------------------------------------------------------------
def find_label(stamps, stamp):
    """test_import.py:14-15 - bisect_left on the (string) stamp column: the first
    ground-truth row whose stamp is >= the frame's (lexicographic on str arrays);
    may return len(stamps), which the caller turns into IndexError."""
    return bisect.bisect_left(stamps, stamp)


def quat_to_transformation(gt7):
    """test_import.py:18-47 - 4x4 from [tx ty tz qx qy qz qw]; q is NOT normalised.
    Arithmetic runs in the dtype of `gt7` (the callers hand in float32 rows,
    test_import.py:199) and is stored into a float64 matrix."""
    t = gt7[0:3]
    qx, qy, qz, qw = gt7[3], gt7[4], gt7[5], gt7[6]
    T = np.zeros((4, 4))
    T[0, 0] = qw * qw + qx * qx - qy * qy - qz * qz
    T[1, 0] = 2 * (qx * qy + qw * qz)
    T[2, 0] = 2 * (qz * qx - qw * qy)
    T[0, 1] = 2 * (qx * qy - qw * qz)
    T[1, 1] = qw * qw - qx * qx + qy * qy - qz * qz
    T[2, 1] = 2 * (qy * qz + qw * qx)
    T[0, 2] = 2 * (qz * qx + qw * qy)
    T[1, 2] = 2 * (qy * qz - qw * qx)
    T[2, 2] = qw * qw - qx * qx - qy * qy + qz * qz
    T[0, 3], T[1, 3], T[2, 3], T[3, 3] = t[0], t[1], t[2], 1
    return T


def inverse_trans(T):
    """test_import.py:80-86 - [R^T, -R^T t; 0 1]."""
    inv = np.zeros((4, 4))
    inv[0:3, 0:3] = T[0:3, 0:3].T
    inv[0:3, 3] = -(inv[0:3, 0:3] @ T[0:3, 3])
    inv[3, 3] = 1
    return inv


def relative_pose(T_old, T_new):
    """test_import.py:207 - inv(T_old) . T_new."""
    return np.dot(inverse_trans(T_old), T_new)


def absolute_position(gt7):
    """slam/preprocess/import_data.py:12-42 - [x, y, z, rotation vector]; the quaternion
    columns (qx, qy, qz, qw) are read as q[0..3] in Euler formulas of a (w, x, y, z)
    quaternion (sic), then the rotation matrix is rebuilt from the angles."""
    g = np.asarray(gt7, dtype=np.float64)
    q0, q1, q2, q3 = g[3], g[4], g[5], g[6]
    phi = np.arctan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2))
    theta = np.arcsin(2 * (q0 * q2 - q3 * q1))
    psi = np.arctan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3))
    ct, st, cp, sp, cs, ss = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi), np.cos(psi), np.sin(psi)
    R = np.array([[ct * cp, ss * st * cp - cs * sp, cs * st * cp + ss * sp],
                  [ct * sp, ss * st * sp + cs * cp, cs * st * sp - ss * cp],
                  [-st, ss * ct, cs * ct]])
    w_abs = np.arccos((np.trace(R) - 1) / 2)
    f = 1 / (2 * np.sin(w_abs))
    w = np.array([f * (R[2, 1] - R[1, 2]) * w_abs, f * (R[0, 2] - R[2, 0]) * w_abs, f * (R[1, 0] - R[0, 1]) * w_abs])
    return np.concatenate((g[0:3], w))


def trans_to_twist(T, clamp=False):
    """test_import.py:50-77 (clamp=False) / model_input.py:46-76 (clamp=True).

    SO(3) log with the reference's quirks kept: the `w * w^T` term is a 1x1
    matrix (|w|^2) that broadcasts over all nine entries of the 3x3; guards
    return zeros(6); an exact identity rotation gives NaN (0 * inf).
    Always returns a flat float64[6] (the reference returns np.matrix[1,6] on
    the main path and ndarray[6] on the guard paths)."""
    R = T[0:3, 0:3]
    arg = (np.trace(R) - 1) / 2
    if arg > 1:
        return np.zeros(6)
    if clamp and np.abs(arg) > 1:
        arg = 1 * arg / np.abs(arg)
    with np.errstate(all="ignore"):
        th = np.arccos(arg)
        s = 1 / (2 * np.sin(th))
        w = np.array([s * (T[2, 1] - T[1, 2]) * th,
                      s * (T[0, 2] - T[2, 0]) * th,
                      s * (T[1, 0] - T[0, 1]) * th])
        w_hat = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
        w_abs = np.linalg.norm(w)
        if w_abs == 0:
            return np.zeros(6)
        ww = float(np.dot(w, w))
        omega = (np.dot(np.eye(3) - R, w_hat) + ww) / (w_abs * w_abs)
        try:
            v = np.dot(np.linalg.inv(omega), T[0:3, 3])
        except np.linalg.LinAlgError:
            v = np.full(3, np.nan)
    return np.concatenate([v, w])


def twist_to_trans(twist, guard=True):
    """test_import.py:89-110 (guard=True) / model_input.py:88-106 (guard=False):
    Rodrigues exp; the translation uses the 3x3 outer product w^T w, so this is
    not the inverse of trans_to_twist (kept as is)."""
    v = np.asarray(twist[0:3], np.float64)
    w = np.asarray(twist[3:6], np.float64)
    w_abs = np.linalg.norm(w)
    w_hat = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if w_abs == 0:
        if guard:
            return np.eye(4)
        # model_input.py:98-100 without the guard: the reference's np.matrix * 0.0
        # goes through BLAS scal, which writes zeros over the NaN of 0/0, so the
        # rotation comes out as I while the translation (an elementwise 0/0) is NaN.
        T = np.eye(4)
        T[0:3, 3] = np.nan
        return T
    with np.errstate(all="ignore"):
        rot = np.eye(3) + w_hat / w_abs * np.sin(w_abs) + np.dot(w_hat, w_hat) / (w_abs * w_abs) * (1 - np.cos(w_abs))
        t = (np.dot(np.dot(np.eye(3) - rot, w_hat), v) + np.dot(np.outer(w, w), v)) / (w_abs * w_abs)
    T = np.zeros((4, 4))
    T[0:3, 0:3] = rot
    T[0:3, 3] = t
    T[3, 3] = 1
    return T


# --------------------------------------------------------------------------
# per-pixel geometry                                                   [pinned]
# --------------------------------------------------------------------------
def point_cloud_dense(depth, K, factor=FACTOR):
    """Dense form of test_import.py:148-155: xyz[H,W,3] float64 and
    valid0[H,W] = all three coordinates != 0 (the compaction rule of :156).
    Operation order as in the reference: z = d/FACTOR; x = ((u-cx)*z)/fx."""
    fx, fy, cx, cy = K
    H, W = depth.shape
    z = depth.astype(np.float64) / np.float64(factor)
    uu = np.arange(W, dtype=np.float64)[None, :] - cx
    vv = np.arange(H, dtype=np.float64)[:, None] - cy
    x = uu * z / fx
    y = vv * z / fy
    xyz = np.stack([x, y, z], axis=-1)
    valid0 = np.all(xyz != 0.0, axis=-1)
    return xyz, valid0


def point_cloud(depth, dataset_int, factor=FACTOR):
    """test_import.py:148-156 - compacted [N_valid,3], row-major pixel order."""
    xyz, valid0 = point_cloud_dense(depth, intrinsics(dataset_int), factor)
    return xyz.reshape(-1, 3)[valid0.reshape(-1)]


def transform_pointcloud(pc, T):
    """test_import.py:159-160 - P.R + t with row vectors (so each point gets
    R^T p + t, kept as is).  Column j: fma(z,R2j, fma(y,R1j, x*R0j)) + t_j - the
    dgemm evaluation order (see fma())."""
    R = T[0:3, 0:3]
    t = T[0:3, 3]
    x, y, z = pc[..., 0], pc[..., 1], pc[..., 2]
    with np.errstate(all="ignore"):
        cols = [fma(z, R[2, j], fma(y, R[1, j], x * R[0, j])) + t[j] for j in range(3)]
    return np.stack(cols, axis=-1)


def backtransform_pointcloud(pc, K):
    """test_import.py:163-168 - (u, v, d) = (x*fx/z + cx, y*fy/z + cy, z);
    multiply, then divide, then add; no rounding, no z>0 test."""
    fx, fy, cx, cy = K
    with np.errstate(all="ignore"):
        d = pc[..., 2]
        u = pc[..., 0] * fx / d + cx
        v = pc[..., 1] * fy / d + cy
    return np.stack([u, v, d], axis=-1)


def inbounds_mask(uvd, dim):
    """Predicate of test_import.py:174: 0 <= u <= dim[0] and 0 <= v <= dim[1]
    (inclusive on both ends; NaN compares False)."""
    with np.errstate(all="ignore"):
        u, v = uvd[..., 0], uvd[..., 1]
        return (0 <= u) & (u <= dim[0]) & (0 <= v) & (v <= dim[1])


def overlap(uvd, dim):
    """test_import.py:171-176 - fraction of the N points that pass the predicate."""
    return float(np.count_nonzero(inbounds_mask(uvd, dim))) / uvd.shape[0]


def overlap_224(uvd):
    """model_input.py:113-119 - same predicate with the window fixed at 224x224."""
    return overlap(uvd, [224.0, 224.0])


def get_mean(images, channels=3):
    """model_input.py:517-522 - per-channel mean over H, then W, then images."""
    acc = np.zeros(channels)
    for im in images:
        acc += np.mean(np.mean(im, axis=0), axis=0)
    return acc / len(images)


def get_data(associations, groundtruth, sequence_length, dynamic_drop=0.0, static_drop=1,
             read_depth=None, rng=np.random):
    """test_import.py:178-246 - relative-twist labels with optional overlap-based
    frame dropping.  `read_depth(path)` replaces the PNG read of :204; the drop
    path uses fr1 intrinsics and a 640x480 window exactly like the reference."""
    associations = associations[0::static_drop]
    rgb_paths = associations[:, 1]
    depth_paths = associations[:, 3]
    n = associations.shape[0]
    twist = np.zeros((n, 6))
    T_old = np.zeros((4, 4))
    drop = 0.0 < dynamic_drop <= 1.0
    deleted = []
    pc_old = None
    ov = None
    for i in range(n):
        row = groundtruth[:, 1:][find_label(groundtruth[:, 0], associations[i, 0])].astype(np.float32)
        T_new = quat_to_transformation(row)
        if drop:
            pc = point_cloud(read_depth(associations[i, 3]), 1)
        if i > 0:
            rel = relative_pose(T_old, T_new)
            if drop:
                back = backtransform_pointcloud(transform_pointcloud(pc_old, rel), intrinsics(1))
                ov = overlap(back, [640, 480])
                if ov < dynamic_drop:
                    twist[i] = trans_to_twist(rel)
                else:
                    deleted.append(i)
            else:
                twist[i] = trans_to_twist(rel)
        if i == 0 or (drop and ov < dynamic_drop):
            T_old = T_new
        if drop:
            pc_old = pc
    if drop:
        twist = np.delete(twist, deleted, 0)
        rgb_paths = np.delete(rgb_paths, deleted, 0)
        depth_paths = np.delete(depth_paths, deleted, 0)
    n = twist.shape[0]
    sequence_length = min(sequence_length, n)
    start = 0
    if n > sequence_length:
        start = rng.randint(0, n - sequence_length)
    sl = slice(start, start + sequence_length)
    return twist[sl], rgb_paths[sl], depth_paths[sl]


def absolute_twist_labels(gt_rows_f32):
    """model_input.py:384-396 - SimpleInputProvider's label precompute: twist of
    the ABSOLUTE pose for i>0 (the previous pose is never used), zeros for i=0."""
    out = np.zeros((len(gt_rows_f32), 6))
    for i, row in enumerate(gt_rows_f32):
        if i > 0:
            out[i] = trans_to_twist(quat_to_transformation(row), clamp=True)
    return out


# --------------------------------------------------------------------------
# losses                                                                [pinned by formula]
# --------------------------------------------------------------------------
def l2_loss_weighted(out, gt, W):
    """cnn_model.py:139-140 - tf.nn.l2_loss((out-gt).W) = 0.5*sum(((out-gt)W)^2)."""
    r = np.dot(np.asarray(out, np.float64) - np.asarray(gt, np.float64), np.asarray(W, np.float64))
    return 0.5 * float(np.sum(r * r))


def lstm_loss(out, gt, W):
    """lstm_model.py:52-55 - sum(((out-gt)W)^2) / batch."""
    r = np.dot(np.asarray(out, np.float64) - np.asarray(gt, np.float64), np.asarray(W, np.float64))
    return float(np.sum(r * r)) / r.shape[0]


def posenet_loss(out7, gt7, beta=1100.0):
    """vgg16_posenet.py:16-25 - sum((x^-x)^2) + beta*sum((q^-q)^2) over the batch."""
    o = np.asarray(out7, np.float64)
    g = np.asarray(gt7, np.float64)
    return float(np.sum((o[:, 0:3] - g[:, 0:3]) ** 2) + beta * np.sum((o[:, 3:7] - g[:, 3:7]) ** 2))


# --------------------------------------------------------------------------
# dense extension: pixel indices, bilinear warp, residual sums        [extension]
# --------------------------------------------------------------------------
def pixel_index(coord):
    """int32 floor of a projected coordinate; saturates at the int32 range;
    NaN -> INT32_MIN."""
    with np.errstate(all="ignore"):
        f = np.floor(coord)
        out = np.clip(np.where(np.isnan(f), float(INT32_MIN), f), float(INT32_MIN), float(INT32_MAX))
    return out.astype(np.int64).astype(np.int32)


def project_dense(depth1, rel, K, factor=FACTOR):
    """Frame-1 pixels -> frame-2 coordinates, dense.  Returns a dict with
    xyz, xyz2 (float64 [H,W,3]), uvd (float64 [H,W,3]), valid0, inb, warp_valid
    (bool [H,W]), iu, iv (int32 [H,W])."""
    H, W = depth1.shape
    xyz, valid0 = point_cloud_dense(depth1, K, factor)
    xyz2 = transform_pointcloud(xyz, rel)
    uvd = backtransform_pointcloud(xyz2, K)
    inb = valid0 & inbounds_mask(uvd, [W, H])
    with np.errstate(all="ignore"):
        u, v, d = uvd[..., 0], uvd[..., 1], uvd[..., 2]
        warp_valid = valid0 & (0 <= u) & (u <= W - 1) & (0 <= v) & (v <= H - 1) & (d > 0)
    return dict(xyz=xyz, xyz2=xyz2, uvd=uvd, valid0=valid0, inb=inb, warp_valid=warp_valid,
                iu=pixel_index(u), iv=pixel_index(v))


def _bilinear(img_f32, y0, y1, x0, x1, fu, fv):
    a = img_f32[y0, x0]
    b = img_f32[y0, x1]
    c = img_f32[y1, x0]
    e = img_f32[y1, x1]
    if a.ndim > fu.ndim:
        fu = fu[..., None]
        fv = fv[..., None]
    top = a + fu * (b - a)
    bot = c + fu * (e - c)
    return top + fv * (bot - top)


def warp_pair(depth1, rgb1, depth2, rgb2, rel, K, factor=FACTOR):
    """Bilinear warp of frame 2 (rgb + depth) to frame-1 pixels and the residuals.

    For warp_valid pixels: taps (x0,y0)=(iu,iv), x1=min(iu+1,W-1), y1=min(iv+1,H-1),
    weights fu=float32(u-floor(u)), fv likewise; interpolation in float32 as
    top=a+fu*(b-a), bot=c+fu*(e-c), val=top+fv*(bot-top).
    warped_depth = bilinear(raw uint16 as float32) / float32(factor).
    geo_valid = warp_valid and all four depth-2 taps != 0.
    r_photo[c] = warped_rgb[c] - rgb1[c]  (float32)
    r_geo      = float32(float64(warped_depth) - d)
    Pixels that are not warp_valid (resp. geo_valid) hold 0 in every output."""
    H, W = depth1.shape
    P = project_dense(depth1, rel, K, factor)
    wv = P["warp_valid"]
    u = np.where(wv, P["uvd"][..., 0], 0.0)
    v = np.where(wv, P["uvd"][..., 1], 0.0)
    x0 = np.floor(u).astype(np.int64)
    y0 = np.floor(v).astype(np.int64)
    x1 = np.minimum(x0 + 1, W - 1)
    y1 = np.minimum(y0 + 1, H - 1)
    fu = (u - np.floor(u)).astype(np.float32)
    fv = (v - np.floor(v)).astype(np.float32)
    rgb2f = rgb2.astype(np.float32)
    d2f = depth2.astype(np.float32)
    w_rgb = _bilinear(rgb2f, y0, y1, x0, x1, fu, fv)
    w_d = _bilinear(d2f, y0, y1, x0, x1, fu, fv) / np.float32(factor)
    taps_ok = (depth2[y0, x0] != 0) & (depth2[y0, x1] != 0) & (depth2[y1, x0] != 0) & (depth2[y1, x1] != 0)
    gv = wv & taps_ok
    r_photo = w_rgb - rgb1.astype(np.float32)
    with np.errstate(all="ignore"):
        r_geo = (w_d.astype(np.float64) - P["uvd"][..., 2]).astype(np.float32)
    w_rgb = np.where(wv[..., None], w_rgb, np.float32(0)).astype(np.float32)
    w_d = np.where(wv, w_d, np.float32(0)).astype(np.float32)
    r_photo = np.where(wv[..., None], r_photo, np.float32(0)).astype(np.float32)
    r_geo = np.where(gv, r_geo, np.float32(0)).astype(np.float32)
    P.update(geo_valid=gv, warped_rgb=w_rgb, warped_depth=w_d, r_photo=r_photo, r_geo=r_geo)
    return P


def pair_sums(P):
    """Reductions of one warped pair: sums float64[4] = (sum|r_photo|, sum r_photo^2,
    sum|r_geo|, sum r_geo^2), counts uint32[4] = (n_valid0, n_inb, n_warp, n_geo)."""
    rp = P["r_photo"].astype(np.float64)
    rg = P["r_geo"].astype(np.float64)
    sums = np.array([np.abs(rp).sum(), (rp * rp).sum(), np.abs(rg).sum(), (rg * rg).sum()])
    counts = np.array([P["valid0"].sum(), P["inb"].sum(), P["warp_valid"].sum(), P["geo_valid"].sum()],
                      dtype=np.uint32)
    return sums, counts


def pair_loss(depth1, rgb1, depth2, rgb2, rel, K, factor=FACTOR):
    """Batched front end: depth1/rgb1 [B,H,W(,3)], depth2/rgb2 [B,k,H,W(,3)] (or
    [B,H,W(,3)] for k=1), rel [B,k,4,4].  Returns sums [B,k,4], counts [B,k,4]."""
    depth2 = depth2 if depth2.ndim == 4 else depth2[:, None]
    rgb2 = rgb2 if rgb2.ndim == 5 else rgb2[:, None]
    rel = rel if rel.ndim == 4 else rel[:, None]
    B, k = rel.shape[0], rel.shape[1]
    sums = np.zeros((B, k, 4))
    counts = np.zeros((B, k, 4), np.uint32)
    for b in range(B):
        for j in range(k):
            P = warp_pair(depth1[b], rgb1[b], depth2[b, j], rgb2[b, j], rel[b, j], K, factor)
            sums[b, j], counts[b, j] = pair_sums(P)
    return sums, counts
