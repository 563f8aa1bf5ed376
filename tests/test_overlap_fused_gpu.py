"""slamhip_overlap_fused (the reference's _point_cloud -> _transform_pointcloud -> _backtransform_pointcloud ->
_overlap chain, test_import.py:148-176, in one pass over the depth plane) against the numpy oracle's chain and
against the materialising kernels: both counts bit for bit."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import slam_oracle as so
from slam_amd import capi
from slam_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fp():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from slam_amd import frame_pairs
    return frame_pairs


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _u32(t):
    return t.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


def _oracle_counts(depth, rel, K, dim, factor):
    """The reference's own chain on one frame: kept points, and those of them inside the window."""
    pc, valid0 = so.point_cloud_dense(depth, K, factor)
    pts = pc.reshape(-1, 3)[valid0.reshape(-1)]
    if pts.shape[0] == 0:
        return 0, 0
    uvd = so.backtransform_pointcloud(so.transform_pointcloud(pts, rel), K)
    return pts.shape[0], int(np.count_nonzero(so.inbounds_mask(uvd, dim)))


def _rels(n, seed, max_deg=3.0, max_t=0.05):
    gt = syn.pose_pairs(n, seed=seed, max_t=max_t, max_deg=max_deg)
    return np.stack([so.relative_pose(so.quat_to_transformation(gt[i, 0]), so.quat_to_transformation(gt[i, 1]))
                     for i in range(n)])


def _check(fp, depth, rel, K, dim, factor):
    d, r = _dev(depth), _dev(rel)
    got = _u32(fp.overlap_fused(d, r, K, dim, factor))       # one launch, counters in a workspace
    want = np.array([_oracle_counts(depth[b], rel[b], K, dim, factor) for b in range(depth.shape[0])], np.int64)
    assert np.array_equal(got, want), (got.tolist(), want.tolist())
    # the entry point without a workspace (clearing kernel + atomics into the output) counts the same
    B, H, W = depth.shape
    lib = capi.load()
    plain = torch.full((B, 2), -1, dtype=torch.int32, device=d.device)
    capi.check(lib.slamhip_overlap_fused(d.data_ptr(), r.data_ptr(), capi.K4(K), float(factor), float(dim[0]), float(dim[1]),
                                         B, H, W, plain.data_ptr(), capi.current_stream()), "slamhip_overlap_fused")
    assert np.array_equal(_u32(plain), want)
    return got


def test_workspace_entry_leaves_its_records_cleared(fp):
    """Calls of different sizes in a row on one workspace: every call leaves the records zero-filled (the workgroup
    that finishes a frame last publishes and clears), so the next call starts from zero without a clearing launch."""
    lib = capi.load()
    H, W = 70, 200
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    ws = torch.zeros((lib.slamhip_overlap_fused_workspace_bytes(9),), dtype=torch.uint8, device="cuda")
    for B, seed in ((9, 1), (2, 2), (5, 3), (9, 4)):
        depth = syn.depth_frames(B, H, W, seed=seed, kind="noise")
        rel = _rels(B, seed=seed, max_t=300.0, max_deg=8.0)
        d, r = _dev(depth), _dev(rel)
        got = torch.full((B, 2), -1, dtype=torch.int32, device="cuda")
        capi.check(lib.slamhip_overlap_fused_ws(d.data_ptr(), r.data_ptr(), capi.K4(K), 1.0, float(W), float(H), B, H, W,
                                                got.data_ptr(), ws.data_ptr(), ws.numel(), capi.current_stream()),
                   "slamhip_overlap_fused_ws")
        torch.cuda.synchronize()
        want = np.array([_oracle_counts(depth[b], rel[b], K, (float(W), float(H)), 1.0) for b in range(B)], np.int64)
        assert np.array_equal(_u32(got), want)
        assert int(ws.to(torch.int64).sum()) == 0
    assert lib.slamhip_overlap_fused_ws(d.data_ptr(), r.data_ptr(), capi.K4(K), 1.0, float(W), float(H), 9, H, W,
                                        got.data_ptr(), ws.data_ptr(), 16, None) < 0          # workspace too small


@pytest.mark.parametrize("H,W,factor,kind", [(5, 7, 1.0, "noise"), (48, 64, 1.0, "plane"), (33, 130, 5000.0, "noise"),
                                             (64, 257, 1.0, "plane"), (120, 160, 5000.0, "plane")])
def test_small_frames_against_the_reference_chain(fp, H, W, factor, kind):
    B = 3
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    depth = syn.depth_frames(B, H, W, seed=H + W, kind=kind)
    depth[0, ::3, ::4] = 0
    rel = _rels(B, seed=H, max_t=0.05 if factor == 5000.0 else 250.0)
    got = _check(fp, depth, rel, K, (float(W), float(H)), factor)
    assert (got[:, 0] > 0).all()
    # a window other than the image (the reference always passes 640 x 480)
    _check(fp, depth, rel, K, (0.6 * W, 0.5 * H), factor)


def test_integer_principal_point_and_degenerate_poses(fp):
    """Points with a zero coordinate are dropped by _point_cloud (col == cx, row == cy, depth 0); identity,
    a flip about x (every z' negative), z' = 0 and a NaN pose take the exact path."""
    H, W = 48, 64
    K = (50.0, 50.0, 32.0, 24.0)
    rs = np.random.RandomState(3)
    depth = rs.randint(1, 3000, size=(5, H, W)).astype(np.uint16)
    depth[:, 5, :] = 0
    rel = np.stack([np.eye(4)] * 5)
    rel[1] = np.diag([1.0, -1.0, -1.0, 1.0])
    rel[2][2, :3] = 0.0                       # z' = t2 = 0 for every point
    rel[3][0, 3] = np.nan
    rel[4][:3, 3] = [40.0, -25.0, 10.0]
    got = _check(fp, depth, rel, K, (float(W), float(H)), 1.0)
    # (under the identity a few border points round to u or v just below 0: the reference drops them, so do we)
    assert got[0, 0] == np.count_nonzero(depth[0][np.arange(H) != 24][:, np.arange(W) != 32]) and got[0, 1] > 0.99 * got[0, 0]
    assert got[3, 1] == 0


def test_projections_within_1e_minus_12_of_the_window_borders(fp):
    """A constant-depth frame under a pure translation t = ((m + d) Z / fx, (n + e) Z / fy, 0) has u = col + m + d,
    v = row + n + e up to rounding: with d, e = +-1e-12 ... 1e-6 whole columns and rows of projections lie that close
    to u = 0, u = dim_u, v = 0 and v = dim_v.  The cheap projection may not decide them."""
    H, W = 48, 64
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    Z = 5000
    offs = [0.0, 1e-12, -1e-12, 1e-10, -1e-10, 1e-9, -1e-9, 1e-8, -1e-7, 1e-6, -1e-6, 3e-6]
    shifts = [(0, 0), (1, 0), (-3, 2), (W - 1, -1), (-(W - 2), -(H - 1)), (2, H - 2)]
    rels = []
    for i, d in enumerate(offs):
        for (m, n) in shifts:
            T = np.eye(4)
            T[0, 3] = (m + d) * Z / K[0]
            T[1, 3] = (n + offs[(i * 5 + 3) % len(offs)]) * Z / K[1]
            rels.append(T)
    rel = np.stack(rels)
    B = rel.shape[0]
    depth = np.full((B, H, W), Z, np.uint16)
    depth[:, ::7, ::5] = 0
    rel5 = rel.copy()
    rel5[:, :3, 3] /= 5000.0      # metric depth: the divisions by the factor are part of the exact path
    for factor, r in ((1.0, rel), (5000.0, rel5)):
        got = _check(fp, depth, r, K, (float(W), float(H)), factor)
        # the construction does what it says: the same shift with offsets of +-1e-12 ... 1e-6 gives different counts
        # (whole columns / rows of projections move across a border)
        assert any(len(set(got[j::len(shifts), 1].tolist())) > 1 for j in range(len(shifts)))


def test_bench_size_against_the_materialising_kernels(fp):
    """64 pairs of 640 x 480: the counts of backproject + reproject (the path _get_data used before), bit for bit,
    and three pairs against the reference chain."""
    B, H, W = 64, 480, 640
    K = fp.intrinsics(1)
    depth = syn.depth_frames(B, H, W, seed=1, kind="plane")
    rel = _rels(B, seed=3)
    d = _dev(depth)
    got = _u32(fp.overlap_fused(d, _dev(rel), K, (640.0, 480.0), 1.0))
    xyz, valid0 = fp.backproject(d, K, 1.0, torch.float64)
    want = _u32(fp.reproject(xyz, _dev(rel), K, (640, 480), valid0=valid0, want=("counts",))["counts"])
    assert np.array_equal(got, want)
    assert (got[:, 1] > 0.5 * got[:, 0]).all()
    for b in (0, 31, 63):
        assert tuple(got[b]) == _oracle_counts(depth[b], rel[b], K, (640.0, 480.0), 1.0)


def test_large_motions_whole_rectangles_inside_outside_and_straddling(fp):
    """Poses that leave whole 64 x 24 rectangles inside the window, whole ones beyond each of its four borders
    and many across them (rotations about the optical axis, sideways and vertical translations, a pitch that
    brings z' near zero for part of the frame), on per-pixel random metric depth (parallax of tens of pixels
    inside one rectangle) and on plane depth: counts of the materialising kernels bit for bit, three pairs
    against the reference chain."""
    H, W = 480, 640
    K = fp.intrinsics(1)

    def rot(axis, deg):
        c, s_ = np.cos(np.deg2rad(deg)), np.sin(np.deg2rad(deg))
        R = np.eye(4)
        i, j = [(1, 2), (0, 2), (0, 1)][axis]
        R[i, i] = c; R[i, j] = -s_; R[j, i] = s_; R[j, j] = c
        return R

    def trans(x, y, z):
        T = np.eye(4)
        T[:3, 3] = [x, y, z]
        return T

    rels = [rot(2, 30.0), rot(2, -75.0), trans(1.2, 0.0, 0.0), trans(-0.9, 0.0, 0.0), trans(0.0, 0.8, 0.0),
            trans(0.0, -1.1, 0.0), trans(0.0, 0.0, -0.45), trans(0.0, 0.0, 2.0), rot(0, 40.0), rot(1, -35.0) @ trans(0.3, 0.1, 0.2),
            rot(2, 180.0), trans(5.0, 0.0, 0.0)]
    rel = np.stack(rels)
    B = rel.shape[0]
    for kind in ("noise", "plane"):
        depth = syn.depth_frames(B, H, W, seed=11, kind=kind)
        depth[1, 100:200, :] = 0
        depth[2, :, 300:310] = 0
        d = _dev(depth)
        got = _u32(fp.overlap_fused(d, _dev(rel), K, (640.0, 480.0), 5000.0))
        xyz, valid0 = fp.backproject(d, K, 5000.0, torch.float64)
        want = _u32(fp.reproject(xyz, _dev(rel), K, (640, 480), valid0=valid0, want=("counts",))["counts"])
        assert np.array_equal(got, want), (kind, got.tolist(), want.tolist())
        frac = got[:, 1] / np.maximum(got[:, 0], 1)
        assert frac.min() < 0.05 and frac.max() > 0.5 and ((frac > 0.1) & (frac < 0.9)).sum() >= 3, frac.tolist()
        for b in (0, 2, 9):
            assert tuple(got[b]) == _oracle_counts(depth[b], rel[b], K, (640.0, 480.0), 5000.0)


def test_argument_errors(fp):
    from slam_amd import capi
    lib = capi.load()
    a16 = torch.zeros(64, dtype=torch.uint8, device="cuda").data_ptr()
    assert lib.slamhip_overlap_fused(a16, a16, capi.K4([500, 500, 64, 12]), 1.0, 640.0, 480.0, 1, 0, 8, a16, None) < 0
    assert lib.slamhip_overlap_fused(None, a16, capi.K4([500, 500, 64, 12]), 1.0, 640.0, 480.0, 1, 4, 8, a16, None) < 0
    assert lib.slamhip_overlap_fused(a16, a16, capi.K4([500, 500, 64, 12]), 1.0, 640.0, 480.0, 70000, 4, 8, a16, None) < 0
    assert lib.slamhip_overlap_fused(None, None, capi.K4([500, 500, 64, 12]), 1.0, 640.0, 480.0, 0, 4, 8, None, None) == 0
    # the one-launch entry keeps a frame's counts and tickets in one 64-bit word: 16384 x 16384 frames do not fit
    # (rejected before anything is launched; the plain entry has no such limit)
    K = capi.K4([500, 500, 64, 12])
    assert lib.slamhip_overlap_fused_ws(a16, a16, K, 1.0, 640.0, 480.0, 1, 16384, 16384, a16, a16, 64, None) == -3
    assert "bits per workspace record" in lib.slamhip_last_error().decode()
