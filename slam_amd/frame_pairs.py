"""Device-resident frame-pair operators (torch tensors in, torch tensors out)
on top of the slamhip C ABI.  torch is plumbing here: it owns the HBM buffers
and the stream; every per-pixel result comes from the HIP kernels.
"""
import numpy as np
import torch

from . import capi

# slam/preprocess/test_import.py:255-263
FX = (517.3, 520.9, 535.4)
FY = (516.5, 521.0, 539.2)
CX = (318.6, 325.1, 320.1)
CY = (255.3, 249.7, 247.6)
FACTOR = 1.0


def intrinsics(dataset_int):
    i = dataset_int - 1
    return (FX[i], FY[i], CX[i], CY[i])


def _dev(t, dtype, what):
    if not (isinstance(t, torch.Tensor) and t.is_cuda):
        raise TypeError("%s must be a CUDA tensor" % what)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (what, dtype, t.dtype))
    return t.contiguous()


def relative_pose(gt7_a, gt7_b):
    """float32 [n,7] ground-truth rows on the device -> rel float64 [n,4,4]
    (inv(T(a)) . T(b), slam/preprocess/test_import.py:199-207)."""
    a = _dev(gt7_a, torch.float32, "gt7_a")
    b = _dev(gt7_b, torch.float32, "gt7_b")
    n = a.shape[0]
    rel = torch.empty((n, 4, 4), dtype=torch.float64, device=a.device)
    lib = capi.load()
    capi.check(lib.slamhip_relative_pose_device(a.data_ptr(), b.data_ptr(), n, rel.data_ptr(),
                                                capi.current_stream()), "slamhip_relative_pose_device")
    return rel


def backproject(depth, K, factor=FACTOR, dtype=torch.float64):
    """depth uint16 [B,H,W] -> (xyz [B,H,W,3] dtype, valid0 uint8 [B,H,W])
    (test_import.py:148-156, dense)."""
    d = _dev(depth, torch.uint16, "depth")
    B, H, W = d.shape
    xyz = torch.empty((B, H, W, 3), dtype=dtype, device=d.device)
    valid0 = torch.empty((B, H, W), dtype=torch.uint8, device=d.device)
    lib = capi.load()
    capi.check(lib.slamhip_backproject(d.data_ptr(), B, H, W, capi.K4(K), float(factor), xyz.data_ptr(),
                                       1 if dtype == torch.float64 else 0, valid0.data_ptr(),
                                       capi.current_stream()), "slamhip_backproject")
    return xyz, valid0


def reproject(xyz, rel, K, dim, valid0=None, dtype=torch.float64, want=("xyz2", "uvd", "iu_iv", "inb", "counts")):
    """xyz [B,H,W,3], rel float64 [B,4,4] or None -> dict of the requested outputs
    (test_import.py:159-176, dense)."""
    if xyz.dtype not in (torch.float64, torch.float32):
        raise TypeError("xyz must be float32/float64")
    x = _dev(xyz, xyz.dtype, "xyz")
    B, H, W, _ = x.shape
    r = None if rel is None else _dev(rel, torch.float64, "rel")
    if r is not None and r.numel() != B * 16:
        raise ValueError("rel must hold one 4x4 per image")
    v0 = None if valid0 is None else _dev(valid0, torch.uint8, "valid0")
    dev = x.device
    out = {}
    if "xyz2" in want:
        out["xyz2"] = torch.empty((B, H, W, 3), dtype=dtype, device=dev)
    if "uvd" in want:
        out["uvd"] = torch.empty((B, H, W, 3), dtype=dtype, device=dev)
    if "iu_iv" in want:
        out["iu_iv"] = torch.empty((B, H, W, 2), dtype=torch.int32, device=dev)
    if "inb" in want:
        out["inb"] = torch.empty((B, H, W), dtype=torch.uint8, device=dev)
    if "counts" in want:
        out["counts"] = torch.empty((B, 2), dtype=torch.int32, device=dev)  # uint32 payload
    lib = capi.load()
    capi.check(lib.slamhip_reproject(x.data_ptr(), 1 if x.dtype == torch.float64 else 0, capi.ptr(v0),
                                     capi.ptr(r), capi.K4(K), float(dim[0]), float(dim[1]), B, H, W,
                                     capi.ptr(out.get("xyz2")), capi.ptr(out.get("uvd")),
                                     1 if dtype == torch.float64 else 0, capi.ptr(out.get("iu_iv")),
                                     capi.ptr(out.get("inb")), capi.ptr(out.get("counts")),
                                     capi.current_stream()), "slamhip_reproject")
    return out


def overlap_count(uvd, dim, want_mask=False):
    """uvd float64 [n,3] -> (count tensor int32[1], mask uint8 [n] or None)
    (test_import.py:171-176)."""
    u = _dev(uvd, torch.float64, "uvd")
    n = u.shape[0]
    cnt = torch.empty((1,), dtype=torch.int32, device=u.device)
    mask = torch.empty((n,), dtype=torch.uint8, device=u.device) if want_mask else None
    lib = capi.load()
    capi.check(lib.slamhip_overlap_count(u.data_ptr(), n, float(dim[0]), float(dim[1]), capi.ptr(mask),
                                         cnt.data_ptr(), capi.current_stream()), "slamhip_overlap_count")
    return cnt, mask


_overlap_ws = {}


def overlap_fused(depth, rel, K, dim, factor=FACTOR):
    """depth uint16 [B,H,W], rel float64 [B,4,4] -> counts int32 [B,2] = {points the reference's _point_cloud
    keeps, of those the ones projected into [0, dim[0]] x [0, dim[1]]} (test_import.py:148-176 in one pass over
    the depth plane; the overlap of :176 is counts[:,1] / counts[:,0])."""
    d = _dev(depth, torch.uint16, "depth")
    if d.dim() != 3:
        raise ValueError("depth must be [B,H,W]")
    B, H, W = d.shape
    r = _dev(rel, torch.float64, "rel")
    if tuple(r.shape) != (B, 4, 4):
        raise ValueError("rel must be [B,4,4]")
    counts = torch.empty((B, 2), dtype=torch.int32, device=d.device)
    lib = capi.load()
    # one launch: the counters' records live in a zero-filled workspace that every call leaves zero-filled (one per
    # device and stream, so that calls on different streams do not share records)
    key = (d.device.index, capi.current_stream())
    ws = _overlap_ws.get(key)
    if ws is None or ws.numel() < 16 * B:
        ws = _overlap_ws[key] = torch.zeros((max(16 * B, 4096),), dtype=torch.uint8, device=d.device)
    try:
        capi.check(lib.slamhip_overlap_fused_ws(d.data_ptr(), r.data_ptr(), capi.K4(K), float(factor), float(dim[0]),
                                                float(dim[1]), B, H, W, counts.data_ptr(), ws.data_ptr(), ws.numel(),
                                                capi.current_stream()), "slamhip_overlap_fused_ws")
    except Exception:
        ws.zero_()
        raise
    return counts


def _pair_args(depth1, rgb1, depth2, rgb2, rel):
    d1 = _dev(depth1, torch.uint16, "depth1")
    c1 = _dev(rgb1, torch.uint8, "rgb1")
    d2 = _dev(depth2, torch.uint16, "depth2")
    c2 = _dev(rgb2, torch.uint8, "rgb2")
    r = _dev(rel, torch.float64, "rel")
    B, H, W = d1.shape
    if d2.dim() == 3:
        d2 = d2.unsqueeze(1)
    if c2.dim() == 4:
        c2 = c2.unsqueeze(1)
    k = d2.shape[1]
    if tuple(c1.shape) != (B, H, W, 3) or tuple(d2.shape) != (B, k, H, W) or tuple(c2.shape) != (B, k, H, W, 3):
        raise ValueError("frame shapes disagree: %s %s %s %s" % (d1.shape, c1.shape, d2.shape, c2.shape))
    if r.numel() != B * k * 16:
        raise ValueError("rel must be [B,k,4,4]")
    return d1, c1, d2.contiguous(), c2.contiguous(), r, B, k, H, W


_workspaces = {}


def _reset_workspace(ws, stream):
    """Zero-fill a workspace on the stream that launches with it."""
    capi.check(capi.load().slamhip_pair_loss_reset_workspace(ws.data_ptr(), ws.numel(), stream),
               "slamhip_pair_loss_reset_workspace")


def _workspace(device, stream, nbytes):
    """Zero-filled scratch of the fused kernel, one per (device, stream); the
    kernel hands it back zero-filled, so it is cleared only when (re)allocated
    (on the launching stream) or after a failed call (`pair_loss` drops it)."""
    key = (device.index, stream)
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty((max(nbytes, 1 << 16),), dtype=torch.uint8, device=device)
        _reset_workspace(ws, stream)
        _workspaces[key] = ws
    return ws


def pair_loss(depth1, rgb1, depth2, rgb2, rel, K, factor=FACTOR, out=None):
    """The fused kernel: returns (sums float64 [B,k,4], counts int32 [B,k,4]).
    sums = (sum|r_photo|, sum r_photo^2, sum|r_geo|, sum r_geo^2),
    counts = (n_valid0, n_inb, n_warp, n_geo).  `out=(sums, counts)` reuses buffers."""
    d1, c1, d2, c2, r, B, k, H, W = _pair_args(depth1, rgb1, depth2, rgb2, rel)
    if out is None:
        sums = torch.empty((B, k, capi.NSUMS), dtype=torch.float64, device=d1.device)
        counts = torch.empty((B, k, capi.NCOUNTS), dtype=torch.int32, device=d1.device)
    else:
        sums, counts = out
    lib = capi.load()
    stream = capi.current_stream()
    ws = _workspace(d1.device, stream, lib.slamhip_pair_loss_workspace_bytes(B, k))
    try:
        capi.check(lib.slamhip_pair_loss_fused(d1.data_ptr(), c1.data_ptr(), d2.data_ptr(), c2.data_ptr(),
                                               r.data_ptr(), capi.K4(K), float(factor), B, k, H, W,
                                               sums.data_ptr(), counts.data_ptr(), ws.data_ptr(), ws.numel(),
                                               stream), "slamhip_pair_loss_fused")
    except Exception:
        # a failed call may leave tickets or partial sums behind: never reuse that buffer as it is
        _workspaces.pop((d1.device.index, stream), None)
        raise
    return sums, counts


class PairLossPlan(object):
    """A prepared call of the fused kernel on fixed buffers: everything that does
    not change between steps (argument checks, ctypes conversion, workspace and
    output allocation) is done once, `run()` is a single C call.  Use it in
    loops that process many batches of one shape; `rebind()` points the plan at
    another set of input tensors of the same shape."""

    def __init__(self, depth1, rgb1, depth2, rgb2, rel, K, factor=FACTOR, stream=None):
        self._lib = capi.load()
        self._fn = self._lib.slamhip_pair_loss_fused
        self.K = capi.K4(K)
        self.factor = float(factor)
        self.stream = capi.current_stream() if stream is None else stream
        self.rebind(depth1, rgb1, depth2, rgb2, rel)
        dev = self._keep[0].device
        self.sums = torch.empty((self.B, self.k, capi.NSUMS), dtype=torch.float64, device=dev)
        self.counts = torch.empty((self.B, self.k, capi.NCOUNTS), dtype=torch.int32, device=dev)
        nbytes = self._lib.slamhip_pair_loss_workspace_bytes(self.B, self.k)
        self._ws = torch.empty((max(nbytes, 64),), dtype=torch.uint8, device=dev)   # owned by this plan
        _reset_workspace(self._ws, self.stream)   # ordered before the launches: same stream
        self._tail = (self.sums.data_ptr(), self.counts.data_ptr(), self._ws.data_ptr(), self._ws.numel(), self.stream)

    def rebind(self, depth1, rgb1, depth2, rgb2, rel):
        d1, c1, d2, c2, r, B, k, H, W = _pair_args(depth1, rgb1, depth2, rgb2, rel)
        if hasattr(self, "B") and (B, k, H, W) != (self.B, self.k, self.H, self.W):
            raise ValueError("rebind needs tensors of the plan's shape")
        self.B, self.k, self.H, self.W = B, k, H, W
        self._keep = (d1, c1, d2, c2, r)
        self._head = (d1.data_ptr(), c1.data_ptr(), d2.data_ptr(), c2.data_ptr(), r.data_ptr())
        return self

    def run(self):
        rc = self._fn(*self._head, self.K, self.factor, self.B, self.k, self.H, self.W, *self._tail)
        if rc != 0:
            try:
                self.reset()
            except RuntimeError:
                pass   # the first failure is the one to report
            capi.check(rc, "slamhip_pair_loss_fused")
        return self.sums, self.counts

    def reset(self):
        """Zero the plan's workspace on its stream (after a failed or interrupted run)."""
        _reset_workspace(self._ws, self.stream)


def warp(depth1, rgb1, depth2, rgb2, rel, K, factor=FACTOR):
    """Per-pixel outputs of the warp (materialising kernel); dict of tensors shaped
    [B,k,H,W(,C)]: warped_rgb, warped_depth, r_photo, r_geo, mask, iu_iv, uvd."""
    d1, c1, d2, c2, r, B, k, H, W = _pair_args(depth1, rgb1, depth2, rgb2, rel)
    dev = d1.device
    o = dict(
        warped_rgb=torch.empty((B, k, H, W, 3), dtype=torch.float32, device=dev),
        warped_depth=torch.empty((B, k, H, W), dtype=torch.float32, device=dev),
        r_photo=torch.empty((B, k, H, W, 3), dtype=torch.float32, device=dev),
        r_geo=torch.empty((B, k, H, W), dtype=torch.float32, device=dev),
        mask=torch.empty((B, k, H, W), dtype=torch.uint8, device=dev),
        iu_iv=torch.empty((B, k, H, W, 2), dtype=torch.int32, device=dev),
        uvd=torch.empty((B, k, H, W, 3), dtype=torch.float64, device=dev),
    )
    lib = capi.load()
    capi.check(lib.slamhip_warp_materialize(d1.data_ptr(), c1.data_ptr(), d2.data_ptr(), c2.data_ptr(),
                                            r.data_ptr(), capi.K4(K), float(factor), B, k, H, W,
                                            o["warped_rgb"].data_ptr(), o["warped_depth"].data_ptr(),
                                            o["r_photo"].data_ptr(), o["r_geo"].data_ptr(),
                                            o["mask"].data_ptr(), o["iu_iv"].data_ptr(), o["uvd"].data_ptr(),
                                            capi.current_stream()), "slamhip_warp_materialize")
    return o


def losses_from_sums(sums, counts):
    """Mean photometric / geometric residuals and the overlap fraction per
    (pair, neighbour): dict of float64 tensors [B,k]."""
    c = counts.to(torch.float64)
    n_valid0, n_inb, n_warp, n_geo = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    return dict(
        photo_l1=sums[..., 0] / (3.0 * n_warp),
        photo_l2=sums[..., 1] / (3.0 * n_warp),
        geo_l1=sums[..., 2] / n_geo,
        geo_l2=sums[..., 3] / n_geo,
        overlap=n_inb / n_valid0,
    )


# ---- host pose maths (numpy in / numpy out) ---------------------------------
def _f32rows(a, width):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1, width))
    return a, a.shape[0]


def quat_to_transformation_batch(gt7):
    g, n = _f32rows(gt7, 7)
    T = np.empty((n, 4, 4))
    capi.check(capi.load().slamhip_quat_to_transformation(g.ctypes.data, n, T.ctypes.data),
               "slamhip_quat_to_transformation")
    return T


def inverse_trans_batch(T):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(-1, 4, 4))
    inv = np.empty_like(T)
    capi.check(capi.load().slamhip_inverse_trans(T.ctypes.data, T.shape[0], inv.ctypes.data),
               "slamhip_inverse_trans")
    return inv


def absolute_position_batch(gt7):
    """[n,7] float64 poses -> [n,6] (slam/preprocess/import_data.py:12-42), on the host."""
    g = np.ascontiguousarray(np.asarray(gt7, dtype=np.float64).reshape(-1, 7))
    out = np.empty((g.shape[0], 6))
    capi.check(capi.load().slamhip_absolute_position(g.ctypes.data, g.shape[0], out.ctypes.data),
               "slamhip_absolute_position")
    return out


def trans_to_twist_batch(T, clamp=False):
    T = np.ascontiguousarray(np.asarray(T, dtype=np.float64).reshape(-1, 4, 4))
    tw = np.empty((T.shape[0], 6))
    capi.check(capi.load().slamhip_trans_to_twist(T.ctypes.data, T.shape[0], int(bool(clamp)), tw.ctypes.data),
               "slamhip_trans_to_twist")
    return tw


def twist_to_trans_batch(tw, guard=True):
    tw = np.ascontiguousarray(np.asarray(tw, dtype=np.float64).reshape(-1, 6))
    T = np.empty((tw.shape[0], 4, 4))
    capi.check(capi.load().slamhip_twist_to_trans(tw.ctypes.data, tw.shape[0], int(bool(guard)), T.ctypes.data),
               "slamhip_twist_to_trans")
    return T


def pose_batch(gt7_a, gt7_b, clamp=False):
    """float32 [n,7] x2 (host) -> (rel float64 [n,4,4], twist float64 [n,6])."""
    a, n = _f32rows(gt7_a, 7)
    b, nb = _f32rows(gt7_b, 7)
    if n != nb:
        raise ValueError("gt7_a and gt7_b differ in length")
    rel = np.empty((n, 4, 4))
    tw = np.empty((n, 6))
    capi.check(capi.load().slamhip_pose_batch(a.ctypes.data, b.ctypes.data, n, int(bool(clamp)),
                                              rel.ctypes.data, tw.ctypes.data), "slamhip_pose_batch")
    return rel, tw
