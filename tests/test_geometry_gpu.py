"""Parity of the HIP frame-pair kernels (through the C ABI) against the numpy
oracle and the golden vectors of the reference.  Integer/byte results (masks,
counts, pixel indices) are compared bit-exactly; float64 coordinates bit-exactly;
float32 intensities, residuals and the reduced sums within 1e-4 relative (the
tolerance the task states for fp32 quantities)."""
import hashlib
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from oracle import slam_oracle as so
from slam_amd.utils import synthetic as syn

pytestmark = pytest.mark.gpu

RTOL = 1e-4


@pytest.fixture(scope="module")
def fp():
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    from slam_amd import frame_pairs
    return frame_pairs


def _sha(a):
    return np.frombuffer(hashlib.sha256(np.ascontiguousarray(a).tobytes()).digest(), np.uint8)


def _dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _u32(t):
    return t.cpu().numpy().astype(np.int64) & 0xFFFFFFFF


def synth_depth(rs, h, w, zero_frac=0.15):
    d = rs.randint(2500, 25001, size=(h, w)).astype(np.uint16)
    d[rs.uniform(size=(h, w)) < zero_frac] = 0
    return d


def test_relative_pose_device_golden(fp, golden_dir):
    g = np.load(os.path.join(golden_dir, "pose.npz"))
    rel = fp.relative_pose(_dev(g["gt_a"]), _dev(g["gt_b"])).cpu().numpy()
    assert np.array_equal(rel, g["rel"])


def test_small_golden_dense(fp, golden_dir):
    g = np.load(os.path.join(golden_dir, "geometry_small.npz"))
    for i in range(int(g["n_cases"])):
        k = "g%d_" % i
        ds = int(g[k + "dataset"])
        K = fp.intrinsics(ds)
        depth, rel, dim = g[k + "depth"], g[k + "rel"], g[k + "dim"]
        xyz, valid0 = fp.backproject(_dev(depth[None]), K)
        v0 = valid0.cpu().numpy().reshape(-1).astype(bool)
        assert np.array_equal(xyz.cpu().numpy().reshape(-1, 3)[v0], g[k + "pc"])
        out = fp.reproject(xyz, _dev(rel[None]), K, dim, valid0=valid0)
        assert np.array_equal(out["xyz2"].cpu().numpy().reshape(-1, 3)[v0], g[k + "tp"])
        assert np.array_equal(out["uvd"].cpu().numpy().reshape(-1, 3)[v0], g[k + "uvd"])
        assert np.array_equal(out["inb"].cpu().numpy().reshape(-1)[v0], g[k + "mask_dim"])
        c = _u32(out["counts"])[0]
        assert c[0] == g[k + "pc"].shape[0]
        assert float(c[1]) / float(c[0]) == float(g[k + "overlap_dim"])
        # float32 outputs are the float64 values rounded once
        xyz32, _ = fp.backproject(_dev(depth[None]), K, dtype=torch.float32)
        assert np.array_equal(xyz32.cpu().numpy(), xyz.cpu().numpy().astype(np.float32))


def test_full_frame_golden(fp, golden_dir):
    g = np.load(os.path.join(golden_dir, "geometry_full.npz"))
    for ds in (1, 2, 3):
        k = "f%d_" % ds
        depth = synth_depth(np.random.RandomState(int(g[k + "seed"])), 480, 640)
        assert np.array_equal(_sha(depth), g[k + "depth_sha256"])
        K = fp.intrinsics(ds)
        xyz, valid0 = fp.backproject(_dev(depth[None]), K)
        out = fp.reproject(xyz, _dev(g[k + "rel"][None]), K, (640, 480), valid0=valid0)
        v0 = valid0.cpu().numpy().reshape(-1).astype(bool)
        pc = xyz.cpu().numpy().reshape(-1, 3)[v0]
        tp = out["xyz2"].cpu().numpy().reshape(-1, 3)[v0]
        uvd = out["uvd"].cpu().numpy().reshape(-1, 3)[v0]
        mask = out["inb"].cpu().numpy().reshape(-1)[v0]
        assert np.array_equal(_sha(pc), g[k + "pc_sha256"])
        assert np.array_equal(_sha(tp), g[k + "tp_sha256"])
        assert np.array_equal(_sha(uvd), g[k + "uvd_sha256"])
        assert np.array_equal(_sha(mask), g[k + "mask_sha256"])
        c = _u32(out["counts"])[0]
        assert c[0] == int(g[k + "n_valid"]) and c[1] == int(g[k + "n_inb"])
        assert float(c[1]) / float(c[0]) == float(g[k + "overlap"])


def test_reference_named_functions(fp, golden_dir):
    from slam_amd.preprocess import geometry as geo
    g = np.load(os.path.join(golden_dir, "geometry_small.npz"))
    for i in (0, 6, 7):
        k = "g%d_" % i
        ds = int(g[k + "dataset"])
        pc = geo._point_cloud(g[k + "depth"], ds)
        assert np.array_equal(pc, g[k + "pc"])
        tp = geo._transform_pointcloud(pc, g[k + "rel"])
        assert np.array_equal(tp, g[k + "tp"])
        uvd = geo._backtransform_pointcloud(tp, ds)
        assert np.array_equal(uvd, g[k + "uvd"])
        assert geo._overlap(uvd, list(g[k + "dim"])) == float(g[k + "overlap_dim"])
        assert geo._overlap(uvd, [640, 480]) == float(g[k + "overlap_640x480"])
    p = np.load(os.path.join(golden_dir, "pose.npz"))
    for i in range(p["gt_a"].shape[0]):
        assert np.array_equal(geo._quat_to_transformation(p["gt_a"][i]), p["T_a"][i])
        tw = geo._trans_to_twist(p["rel"][i])
        assert (tw.ndim, tw.size) == tuple(p["twist_ndim_size"][i])
        np.testing.assert_allclose(np.asarray(tw).reshape(6), p["twist"][i], rtol=1e-11, atol=1e-13, equal_nan=True)


def test_get_data_facade(fp, golden_dir):
    from slam_amd.preprocess import geometry as geo
    g = np.load(os.path.join(golden_dir, "get_data.npz"))
    assoc, gt, depths = g["assoc"], g["gt"], g["depths"]
    index = {assoc[i, 3]: i for i in range(assoc.shape[0])}
    tw, rp, dp = geo._get_data(assoc, gt, 1000)
    np.testing.assert_allclose(tw, g["a_twist"], rtol=1e-11, atol=1e-13)
    assert list(rp) == list(g["a_rgb"]) and list(dp) == list(g["a_depth"])
    tw, rp, dp = geo._get_data(assoc, gt, 4, 0.0, 2, rng=np.random.RandomState(int(g["b_seed"])))
    np.testing.assert_allclose(tw, g["b_twist"], rtol=1e-11, atol=1e-13)
    assert list(rp) == list(g["b_rgb"])
    for tag in ("c", "d"):
        tw, rp, dp = geo._get_data(assoc, gt, 1000, float(g[tag + "_thr"]), 1,
                                   read_depth=lambda p: depths[index[p]])
        np.testing.assert_allclose(tw, g[tag + "_twist"], rtol=1e-11, atol=1e-13)
        assert list(rp) == list(g[tag + "_rgb"]) and list(dp) == list(g[tag + "_depth"])


def _pairs(B, k, H, W, seed, depth_kind="plane", rgb_kind="smooth", max_deg=3.0, max_t=0.05):
    gt = syn.pose_pairs(B * k, seed=seed, max_t=max_t, max_deg=max_deg)
    d1 = syn.depth_frames(B, H, W, seed=seed, kind=depth_kind)
    c1 = syn.rgb_frames(B, H, W, seed=seed, kind=rgb_kind)
    d2 = syn.depth_frames(B * k, H, W, seed=seed + 1, kind=depth_kind).reshape(B, k, H, W)
    c2 = syn.rgb_frames(B * k, H, W, seed=seed + 1, kind=rgb_kind).reshape(B, k, H, W, 3)
    rel = np.stack([so.relative_pose(so.quat_to_transformation(gt[i, 0]), so.quat_to_transformation(gt[i, 1]))
                    for i in range(B * k)]).reshape(B, k, 4, 4)
    return d1, c1, d2, c2, rel


def _check_warp(fp, d1, c1, d2, c2, rel, K, factor):
    o = fp.warp(_dev(d1), _dev(c1), _dev(d2), _dev(c2), _dev(rel), K, factor)
    o = {n: t.cpu().numpy() for n, t in o.items()}
    B, k = rel.shape[0], rel.shape[1]
    for b in range(B):
        for j in range(k):
            P = so.warp_pair(d1[b], c1[b], d2[b, j], c2[b, j], rel[b, j], K, factor)
            m = o["mask"][b, j]
            assert np.array_equal((m & 1) != 0, P["valid0"])
            assert np.array_equal((m & 2) != 0, P["inb"])
            assert np.array_equal((m & 4) != 0, P["warp_valid"])
            assert np.array_equal((m & 8) != 0, P["geo_valid"])
            assert np.array_equal(o["iu_iv"][b, j, ..., 0], P["iu"])
            assert np.array_equal(o["iu_iv"][b, j, ..., 1], P["iv"])
            assert np.array_equal(o["uvd"][b, j], P["uvd"], equal_nan=True)
            np.testing.assert_allclose(o["warped_rgb"][b, j], P["warped_rgb"], rtol=RTOL, atol=1e-4)
            np.testing.assert_allclose(o["warped_depth"][b, j], P["warped_depth"], rtol=RTOL, atol=1e-4)
            np.testing.assert_allclose(o["r_photo"][b, j], P["r_photo"], rtol=RTOL, atol=1e-3)
            np.testing.assert_allclose(o["r_geo"][b, j], P["r_geo"], rtol=RTOL, atol=1e-2)
    return o


def _check_fused(fp, d1, c1, d2, c2, rel, K, factor):
    sums, counts = fp.pair_loss(_dev(d1), _dev(c1), _dev(d2), _dev(c2), _dev(rel), K, factor)
    want_s, want_c = so.pair_loss(d1, c1, d2, c2, rel, K, factor)
    assert np.array_equal(_u32(counts), want_c.astype(np.int64))
    np.testing.assert_allclose(sums.cpu().numpy(), want_s, rtol=RTOL, atol=1e-6)
    return sums, counts


@pytest.mark.parametrize("H,W,factor,dk,ck", [
    (48, 64, 1.0, "noise", "noise"),
    (48, 64, 5000.0, "plane", "smooth"),
    (37, 53, 1.0, "plane", "noise"),       # ragged: W not a multiple of 8, partial tiles
    (16, 128, 1.0, "plane", "smooth"),     # exactly one tile
    (5, 7, 1.0, "noise", "noise"),
])
def test_warp_and_fused_small(fp, H, W, factor, dk, ck):
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    d1, c1, d2, c2, rel = _pairs(3, 2, H, W, seed=H * 1000 + W, depth_kind=dk, rgb_kind=ck)
    _check_warp(fp, d1, c1, d2, c2, rel, K, factor)
    _check_fused(fp, d1, c1, d2, c2, rel, K, factor)


@pytest.mark.parametrize("B,k", [(1, 3), (9, 5), (11, 7), (2, 16)])
def test_fused_neighbour_counts_and_ragged_batches(fp, B, k):
    """Neighbour counts that are no powers of two and batches that are no multiples of eight (the grid deals
    groups of eight anchors to the XCDs and finds the neighbour of a workgroup by a reciprocal multiplication)."""
    H, W = 50, 136      # two tile columns, three tile rows of 24, ragged on both sides
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    d1, c1, d2, c2, rel = _pairs(B, k, H, W, seed=B * 100 + k, depth_kind="plane", rgb_kind="noise")
    _check_fused(fp, d1, c1, d2, c2, rel, K, 1.0)


@pytest.mark.parametrize("seed", range(12))
def test_fused_random_shapes(fp, seed):
    """Seeded random draws of every launch parameter at once: batch, neighbour count, frame size (partial
    tiles on either side), depth factor, scene kind, rotation and translation range."""
    rs = np.random.RandomState(9000 + seed)
    B, k = int(rs.randint(1, 12)), int(rs.randint(1, 7))
    H, W = int(rs.randint(3, 80)), int(rs.randint(3, 300))
    factor = float(rs.choice([1.0, 1000.0, 5000.0]))
    dk, ck = str(rs.choice(["plane", "noise"])), str(rs.choice(["smooth", "noise"]))
    deg, t = float(rs.choice([0.5, 3.0, 25.0])), float(rs.choice([0.0, 0.05, 0.5]))
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    d1, c1, d2, c2, rel = _pairs(B, k, H, W, seed=seed + 77, depth_kind=dk, rgb_kind=ck, max_deg=deg, max_t=t)
    _check_fused(fp, d1, c1, d2, c2, rel, K, factor)
    if B * k <= 12:
        _check_warp(fp, d1, c1, d2, c2, rel, K, factor)


@pytest.mark.parametrize("factor,dk,ck,deg", [
    (1.0, "plane", "smooth", 3.0),
    (1.0, "noise", "noise", 3.0),
    (5000.0, "plane", "smooth", 3.0),      # metric depth: parallax, LDS window per tile
    (5000.0, "noise", "noise", 3.0),       # per-pixel random depth: scattered taps, fallback path
    (1.0, "plane", "noise", 40.0),         # large rotation: most points leave the image
])
def test_warp_and_fused_full_frame(fp, factor, dk, ck, deg):
    K = fp.intrinsics(1)
    d1, c1, d2, c2, rel = _pairs(2, 1, 480, 640, seed=int(deg * 7 + factor), depth_kind=dk, rgb_kind=ck, max_deg=deg)
    _check_warp(fp, d1, c1, d2, c2, rel, K, factor)
    _check_fused(fp, d1, c1, d2, c2, rel, K, factor)


def test_edge_cases(fp):
    K = fp.intrinsics(1)
    H, W = 32, 64
    d1, c1, d2, c2, rel = _pairs(2, 1, H, W, seed=5)
    # all-zero depth: nothing valid
    z = np.zeros_like(d1)
    s, c = _check_fused(fp, z, c1, d2, c2, rel, K, 1.0)
    assert not _u32(c).any() and not s.cpu().numpy().any()
    # identity pose
    eye = np.tile(np.eye(4), (2, 1, 1, 1))
    _check_warp(fp, d1, c1, d2, c2, eye, K, 1.0)
    _check_fused(fp, d1, c1, d2, c2, eye, K, 1.0)
    # pose that puts every point behind the camera / at z' == 0 (inf, NaN coordinates)
    flip = eye.copy()
    flip[:, :, 2, 2] = -1.0
    _check_warp(fp, d1, c1, d2, c2, flip, K, 1.0)
    _check_fused(fp, d1, c1, d2, c2, flip, K, 1.0)
    zero = eye.copy()
    zero[:, :, 2, 2] = 0.0
    _check_warp(fp, d1, c1, d2, c2, zero, K, 1.0)
    _check_fused(fp, d1, c1, d2, c2, zero, K, 1.0)
    # maximum depth values
    full = np.full_like(d1, 65535)
    _check_fused(fp, full, c1, full[:, None], c2, rel, K, 1.0)
    # empty batch
    s, c = fp.pair_loss(_dev(d1[:0]), _dev(c1[:0]), _dev(d2[:0]), _dev(c2[:0]), _dev(rel[:0]), K, 1.0)
    assert s.shape[0] == 0 and c.shape[0] == 0


def test_fused_equals_materialised_at_bench_size(fp):
    """Batch 64 at 640x480 (the benchmark configuration): the oracle would take
    minutes, so check the size-independent property that the fused reduction
    equals the reduction of the materialised per-pixel outputs, counts exactly."""
    K = fp.intrinsics(1)
    B = 64
    d1 = _dev(syn.depth_frames(B, 480, 640, seed=1, kind="plane"))
    d2 = _dev(syn.depth_frames(B, 480, 640, seed=2, kind="plane"))
    c1 = _dev(syn.rgb_frames(B, 480, 640, seed=1, kind="noise"))
    c2 = _dev(syn.rgb_frames(B, 480, 640, seed=2, kind="noise"))
    gt = _dev(syn.pose_pairs(B, seed=3))
    rel = fp.relative_pose(gt[:, 0].contiguous(), gt[:, 1].contiguous()).reshape(B, 1, 4, 4)
    sums, counts = fp.pair_loss(d1, c1, d2, c2, rel, K, 1.0)
    cnt = np.zeros((B, 4), np.int64)
    ref = np.zeros((B, 4))
    for lo in range(0, B, 8):
        sl = slice(lo, lo + 8)
        o = fp.warp(d1[sl], c1[sl], d2[sl], c2[sl], rel[sl], K, 1.0)
        m = o["mask"][:, 0]
        for q in range(4):
            cnt[sl, q] = ((m >> q) & 1).sum(dim=(1, 2)).cpu().numpy()
        rp = o["r_photo"][:, 0].double()
        rg = o["r_geo"][:, 0].double()
        ref[sl, 0] = rp.abs().sum(dim=(1, 2, 3)).cpu().numpy()
        ref[sl, 1] = (rp * rp).sum(dim=(1, 2, 3)).cpu().numpy()
        ref[sl, 2] = rg.abs().sum(dim=(1, 2)).cpu().numpy()
        ref[sl, 3] = (rg * rg).sum(dim=(1, 2)).cpu().numpy()
    assert np.array_equal(_u32(counts)[:, 0], cnt)
    np.testing.assert_allclose(sums.cpu().numpy()[:, 0], ref, rtol=1e-5)
    assert (cnt[:, 2] > 0.5 * 480 * 640).all()  # TUM-like motion keeps most pixels in view


def test_both_tile_shapes_of_the_single_neighbour_kernel(fp):
    """Single-neighbour launches take 128 x 16 tiles up to eight rounds of workgroups and 128 x 24 tiles beyond:
    960 pairs of 72 x 384 in one call (128 x 24) against the same pairs in calls of 240 (128 x 16), counts bit for
    bit, and the first and last pairs against the oracle."""
    B, H, W = 960, 72, 384
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    assert B * 3 * 3 > 8 * 1024 >= 240 * 3 * 5      # workgroups of 128 x 24 tiles in the large and the small calls
    d1, c1, d2, c2, rel = _pairs(B, 1, H, W, seed=21, depth_kind="noise", rgb_kind="noise")
    d1[3, ::4, ::3] = 0
    D = [_dev(x) for x in (d1, c1, d2, c2, rel)]
    big_s, big_c = fp.pair_loss(*D, K, 1.0)
    for lo in range(0, B, 240):
        sl = slice(lo, lo + 240)
        s, c = fp.pair_loss(*[x[sl].contiguous() for x in D], K, 1.0)
        assert torch.equal(c, big_c[sl])
        np.testing.assert_allclose(s.cpu().numpy(), big_s[sl].cpu().numpy(), rtol=2e-6)   # (a wave adds its lanes in float32)
    for sl in (slice(0, 4), slice(B - 4, B)):
        want_s, want_c = so.pair_loss(d1[sl], c1[sl], d2[sl], c2[sl], rel[sl], K, 1.0)
        assert np.array_equal(_u32(big_c[sl]), want_c.astype(np.int64))
        np.testing.assert_allclose(big_s[sl].cpu().numpy(), want_s, rtol=RTOL, atol=1e-6)


def test_high_res_four_neighbours(fp):
    """1280x960 frames, k = 4 neighbours per anchor, intrinsics doubled (the high-resolution
    configuration): counts and sums of the fused kernel against the materialised per-pixel
    outputs for every neighbour, and neighbour 2 of anchor 0 against the C oracle."""
    from oracle import c_oracle
    H, W, B, k = 960, 1280, 2, 4
    K = tuple(2.0 * v for v in fp.intrinsics(1))
    d1, c1, d2, c2, rel = _pairs(B, k, H, W, seed=77, depth_kind="plane", rgb_kind="smooth")
    D1, C1, D2, C2, R = _dev(d1), _dev(c1), _dev(d2), _dev(c2), _dev(rel)
    sums, counts = fp.pair_loss(D1, C1, D2, C2, R, K, 1.0)
    o = fp.warp(D1, C1, D2, C2, R, K, 1.0)
    got_c = _u32(counts)
    for q in range(4):
        want = ((o["mask"] >> q) & 1).sum(dim=(2, 3)).cpu().numpy()
        assert np.array_equal(got_c[..., q], want)
    rp, rg = o["r_photo"].double(), o["r_geo"].double()
    ref = np.stack([rp.abs().sum(dim=(2, 3, 4)).cpu().numpy(), (rp * rp).sum(dim=(2, 3, 4)).cpu().numpy(),
                    rg.abs().sum(dim=(2, 3)).cpu().numpy(), (rg * rg).sum(dim=(2, 3)).cpu().numpy()], axis=-1)
    np.testing.assert_allclose(sums.cpu().numpy(), ref, rtol=RTOL)
    s_o, c_o = c_oracle.pair_loss(d1[:1], c1[:1], d2[:1, 2:3], c2[:1, 2:3], rel[:1, 2:3], K, 1.0)
    assert np.array_equal(got_c[0, 2], c_o[0, 0].astype(np.int64))
    np.testing.assert_allclose(sums.cpu().numpy()[0, 2], s_o[0, 0], rtol=RTOL)


def test_workspace_recovery(fp):
    """A workspace left dirty by an interrupted call gives wrong sums silently; the reset
    entry point (and the plan's reset()) bring it back."""
    K = fp.intrinsics(1)
    d1, c1, d2, c2, rel = _pairs(3, 1, 48, 64, seed=9)
    args = [_dev(a) for a in (d1, c1, d2, c2, rel)]
    want_s, want_c = fp.pair_loss(*args, K, 1.0)
    plan = fp.PairLossPlan(*args, K, 1.0)
    s, c = plan.run()
    torch.cuda.synchronize()
    assert np.array_equal(_u32(c), _u32(want_c))
    plan._ws.fill_(0x5A)            # what an aborted launch may leave: tickets and partial sums
    plan.reset()
    s, c = plan.run()
    torch.cuda.synchronize()
    assert np.array_equal(_u32(c), _u32(want_c))
    np.testing.assert_allclose(s.cpu().numpy(), want_s.cpu().numpy(), rtol=1e-12)


def test_decision_boundaries_adversarial(fp):
    """Projected coordinates placed within 1e-12 .. 1e-6 pixels of integers (and so of every decision the
    reference takes: 0, W-1, W, H-1, H and the floor that picks the taps): a constant-depth anchor under a
    pure translation t = ((m + d) Z / fx, (n + e) Z / fy, 0) has u = col + m + d, v = row + n + e up to
    rounding.  The fused kernel may not certify such pixels with its cheap projection: masks, counts and
    tap choice (n_geo, the sums) must still be the oracle's, bit for bit / within tolerance."""
    H, W = 48, 64
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    Z = 5000
    offs = [0.0, 1e-12, -1e-12, 1e-10, -1e-10, 1e-9, -1e-9, 1e-8, -1e-7, 1e-6, -1e-6, 3e-6]
    shifts = [(0, 0), (3, -2), (-5, 4), (W - 2, 1), (-(W - 3), -(H - 2)), (1, H - 1)]
    rels, B = [], 0
    for i, d in enumerate(offs):
        for (m, n) in shifts[: 3 if i > 6 else 6]:
            e = offs[(i * 5 + 3) % len(offs)]
            T = np.eye(4)
            T[0, 3] = (m + d) * Z / K[0]
            T[1, 3] = (n + e) * Z / K[1]
            rels.append(T)
            B += 1
    rel = np.stack(rels).reshape(B, 1, 4, 4)
    rs = np.random.RandomState(7)
    d1 = np.full((B, H, W), Z, np.uint16)
    d1[:, ::7, ::5] = 0                                   # some invalid anchors
    d1[B // 2:, 1::2, :] = 2 * Z                          # a second depth: half of those rows land mid-pixel
    d2 = rs.randint(2500, 25001, size=(B, 1, H, W)).astype(np.uint16)
    d2[rs.uniform(size=d2.shape) < 0.1] = 0
    c1 = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    c2 = rs.randint(0, 256, size=(B, 1, H, W, 3)).astype(np.uint8)
    for factor in (1.0,):
        o = _check_warp(fp, d1, c1, d2, c2, rel, K, factor)
        _check_fused(fp, d1, c1, d2, c2, rel, K, factor)
    # the construction does what it says: most valid pixels sit within 1e-5 of an integer coordinate
    u = o["uvd"][..., 0]
    near = np.abs(u - np.round(u)) < 1e-5
    assert near[np.isfinite(u)].mean() > 0.5
    # metric depth as well (the divisions by the factor are part of the exact path)
    rel5 = rel.copy()
    rel5[:, :, :3, 3] /= 5000.0
    _check_warp(fp, d1, c1, d2, c2, rel5, K, 5000.0)
    _check_fused(fp, d1, c1, d2, c2, rel5, K, 5000.0)


@pytest.mark.parametrize("factor", [1.0, 5000.0])
def test_coordinates_just_below_zero_beside_the_far_border(fp, factor):
    """Pixels that project to u in (-1, 0) while v lies in the last row [H-1, H) (and the mirror image: v in (-1, 0),
    u in the last column), placed inside tiles whose other pixels project well inside the image, so that the tile's
    tap window does not start at the border.  Such a pixel fails `0 <= u` (test_import.py:174); a floor taken by
    truncation would give index 0 for it.  One target pixel per column (row) under a pure translation:
    u = col + tx fx / z with the integer raw depth z nearest to fx |tx| / (col + 0.5)."""
    H, W = 96, 256
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    fx, fy = K[0], K[1]
    rs = np.random.RandomState(3)
    B = 4
    d1 = np.full((B, H, W), 60000, np.uint16)
    rel = np.tile(np.eye(4), (B, 1, 1, 1))
    hits = 0
    for b in range(B):
        mirror = b % 2 == 1
        t_main, t_side = 3000.0 + 100.0 * b, 60.0 + 7.0 * b
        if not mirror:      # u just below 0, v in the last row
            rel[b, 0, 0, 3], rel[b, 0, 1, 3] = -t_main, t_side
            for c in range(24, W):
                z = int(round(fx * t_main / (c + 0.5)))
                shift_v = t_side * fy / z
                row = H - 1 - int(np.floor(shift_v))
                u, v = c - t_main * fx / z, row + shift_v
                if 1 <= z <= 65535 and 0 <= row < H and -1 < u < 0 and H - 1 <= v < H:
                    d1[b, row, c] = z
                    hits += 1
        else:               # v just below 0, u in the last column
            rel[b, 0, 1, 3], rel[b, 0, 0, 3] = -t_main / 4, t_side
            for r in range(12, H):
                z = int(round(fy * t_main / 4 / (r + 0.5)))
                shift_u = t_side * fx / z
                col = W - 1 - int(np.floor(shift_u))
                u, v = col + shift_u, r - t_main / 4 * fy / z
                if 1 <= z <= 65535 and 0 <= col < W and -1 < v < 0 and W - 1 <= u < W:
                    d1[b, r, col] = z
                    hits += 1
    assert hits > 300
    rel[:, :, :3, 3] /= factor
    d1[:, ::9, ::11] = 0
    d2 = rs.randint(2500, 25001, size=(B, 1, H, W)).astype(np.uint16)
    c1 = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    c2 = rs.randint(0, 256, size=(B, 1, H, W, 3)).astype(np.uint8)
    o = _check_warp(fp, d1, c1, d2, c2, rel, K, factor)
    # the construction does what it says, in the oracle's own coordinates
    u, v = o["uvd"][..., 0], o["uvd"][..., 1]
    corner = ((u > -1) & (u < 0) & (v >= H - 1) & (v < H)) | ((v > -1) & (v < 0) & (u >= W - 1) & (u < W))
    assert int(corner.sum()) > 250
    _check_fused(fp, d1, c1, d2, c2, rel, K, factor)


def test_near_pixels_under_a_small_translation_project_just_above_the_image(fp):
    """A translation so small that a tile's tap window does not depend on depth (it is placed from the tile's
    corners, two rows above the tile) while a few very near pixels (raw depth 2 or 3) still move by r + 0.5 rows:
    a row r of the second tile row lands at v = -0.5, and its last pixel at u in [W-1, W).  That pixel is outside
    the image (`0 <= v` fails, test_import.py:174) although the truncated window-relative row index is 0."""
    H, W = 96, 128
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    fx, fy = K[0], K[1]
    rs = np.random.RandomState(4)
    # (the window is placed without the depths while max(fx, fy) (|tx| + |ty| + |tz|) / 256 < 0.5)
    cases = [(r, z) for r in range(16, 48) for z in (2, 3) if ((r + 0.5) / fy + 0.4 / fx) * z * max(fx, fy) / 256.0 < 0.45]
    B = len(cases)
    assert B >= 20
    d1 = rs.randint(20000, 60000, size=(B, H, W)).astype(np.uint16)
    rel = np.tile(np.eye(4), (B, 1, 1, 1))
    for b, (r, z) in enumerate(cases):
        rel[b, 0, 1, 3] = -(r + 0.5) * z / fy        # v = row - (r + 0.5) at depth z
        rel[b, 0, 0, 3] = 0.4 * z / fx               # u = col + 0.4
        d1[b, r, :] = z
    d2 = rs.randint(2500, 25001, size=(B, 1, H, W)).astype(np.uint16)
    c1 = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    c2 = rs.randint(0, 256, size=(B, 1, H, W, 3)).astype(np.uint8)
    o = _check_warp(fp, d1, c1, d2, c2, rel, K, 1.0)
    u, v = o["uvd"][:, 0, ..., 0], o["uvd"][:, 0, ..., 1]
    for b, (r, z) in enumerate(cases):
        assert -1 < v[b, r, W - 1] < 0 and W - 1 <= u[b, r, W - 1] < W
    _check_fused(fp, d1, c1, d2, c2, rel, K, 1.0)


def test_decision_boundaries_adversarial_under_rotation(fp):
    """One target pixel per pair whose projection under a ROTATION of 1 .. 35 degrees (plus translation) lies within
    1e-12 .. 1e-6 of a decision of the reference - the borders 0, W-1, W / 0, H-1, H (test_import.py:174 and the tap
    rule) or an interior integer (the floor that picks the taps): the translation is solved for, then refined on the
    oracle's own float64 result, so that the target is met in the reference's arithmetic.  The certificate's bound
    depends on the rotation, the depth and the intrinsics; a cheap projection that certified such a pixel would move
    a count.  Masks, pixel indices and counts bit for bit."""
    H, W = 24, 40
    rs = np.random.RandomState(21)
    offs = [0.0, 1e-12, -1e-12, 1e-10, -1e-9, 1e-9, 1e-8, -1e-7, 1e-6, -1e-6]
    for factor in (1.0, 5000.0):
        for K in ((517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480), (70.3, 69.1, 19.5, 12.25)):
            B = 48
            d1 = rs.randint(2000, 60001, size=(B, H, W)).astype(np.uint16)
            d1[rs.uniform(size=d1.shape) < 0.05] = 0
            rel = np.tile(np.eye(4), (B, 1, 1, 1))
            hit = 0
            for b in range(B):
                axis = rs.normal(size=3)
                axis /= np.linalg.norm(axis)
                ang = np.deg2rad(rs.uniform(1.0, 35.0))
                Kx = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
                R = np.eye(3) + np.sin(ang) * Kx + (1 - np.cos(ang)) * Kx @ Kx
                c, r = int(rs.randint(2, W - 2)), int(rs.randint(2, H - 2))
                D = int(rs.randint(2000, 60001))
                d1[b, r, c] = D
                z = D / factor
                p = np.array([(c - K[2]) * z / K[0], (r - K[3]) * z / K[1], z])
                u_t = float(rs.choice([0, W - 1, W, int(rs.randint(1, W - 1))])) + offs[b % len(offs)]
                v_t = float(rs.choice([0, H - 1, H, int(rs.randint(1, H - 1))])) + offs[(b * 3 + 1) % len(offs)]
                if b % 2:
                    u_t = float(int(rs.randint(1, W - 1))) + 0.37      # one coordinate adversarial at a time
                else:
                    v_t = float(int(rs.randint(1, H - 1))) + 0.61
                t = np.zeros(3)
                t[2] = rs.uniform(-0.2, 0.2) * z
                zp = p @ R[:, 2] + t[2]
                if zp <= 0.1 * z:
                    continue
                t[0] = (u_t - K[2]) * zp / K[0] - p @ R[:, 0]
                t[1] = (v_t - K[3]) * zp / K[1] - p @ R[:, 1]
                T = np.eye(4)
                T[:3, :3], T[:3, 3] = R, t
                for _ in range(4):      # refine on the oracle's own result at the target pixel
                    P = so.warp_pair(d1[b], np.zeros((H, W, 3), np.uint8), d1[b], np.zeros((H, W, 3), np.uint8), T, K, factor)
                    u, v, d = P["uvd"][r, c]
                    T[0, 3] += (u_t - u) * d / K[0]
                    T[1, 3] += (v_t - v) * d / K[1]
                P = so.warp_pair(d1[b], np.zeros((H, W, 3), np.uint8), d1[b], np.zeros((H, W, 3), np.uint8), T, K, factor)
                u, v, _ = P["uvd"][r, c]
                hit += int(abs(u - u_t) < 2e-9 and abs(v - v_t) < 2e-9)
                rel[b, 0] = T
            assert hit > 0.8 * B, hit      # the construction does what it says
            d2 = rs.randint(2500, 25001, size=(B, 1, H, W)).astype(np.uint16)
            d2[rs.uniform(size=d2.shape) < 0.1] = 0
            c1 = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
            c2 = rs.randint(0, 256, size=(B, 1, H, W, 3)).astype(np.uint8)
            _check_warp(fp, d1, c1, d2, c2, rel, K, factor)
            _check_fused(fp, d1, c1, d2, c2, rel, K, factor)


@pytest.mark.parametrize("direction", [+1, -1])
def test_taps_gathered_at_the_end_of_the_neighbour_planes(fp, direction):
    """Certified pixels whose taps lie outside the staged window are gathered with vector loads that reach past the
    taps' own bytes; at the last pixels of a plane those loads are pulled back to stay inside it.  Few pixels carry
    depth here, so that a single wrong tap or anchor colour moves the sums far beyond the tolerance: one row at 5 m
    (it lays the window next to the tile) and one near row whose depths send every pixel to the same far column -
    +1: taps (W - 2, H - 2) .. (W - 1, H - 1), the last pixel of the neighbour's planes; -1: anchors in the last
    row up to the anchor planes' last pixel, moving up and left across the image."""
    H, W, factor = 32, 320, 5000.0
    K = (517.3 * W / 640, 516.5 * H / 480, 318.6 * W / 640, 255.3 * H / 480)
    B = 3
    rs = np.random.RandomState(11 + direction)
    c1 = rs.randint(0, 256, size=(B, H, W, 3)).astype(np.uint8)
    c2 = rs.randint(0, 256, size=(B, 1, H, W, 3)).astype(np.uint8)
    d2 = rs.randint(2500, 25001, size=(B, 1, H, W)).astype(np.uint16)
    d1 = np.zeros((B, H, W), np.uint16)
    rel = np.tile(np.eye(4), (B, 1, 1, 1))
    if direction > 0:
        tx = 0.5
        rel[:, :, 0, 3], rel[:, :, 1, 3] = tx, 0.004             # (v = row + 0.2 .. 0.35: no integer, so certified)
        cols = np.arange(0, 128)
        d1[:, 20, cols] = 25000
        z = K[0] * tx / ((W - 1.5) - cols)                      # u = col + fx tx / z = W - 1.5
        d1[:, H - 2, cols] = np.round(z * factor).astype(np.uint16)
    else:
        tx, ty = -0.5, -0.1
        rel[:, :, 0, 3], rel[:, :, 1, 3] = tx, ty
        cols = np.arange(W - 64, W)
        d1[:, 20, cols] = 25000
        d1[:, H - 1, cols] = 2250 + 40 * np.arange(64)           # 0.45 .. 0.95 m: some 130 .. 290 columns to the left
    want_s, want_c = so.pair_loss(d1, c1, d2, c2, rel, K, factor)
    assert want_c[..., 2].min() >= 64 + 30                       # the near rows do warp (the far row always does)
    sums, counts = fp.pair_loss(_dev(d1), _dev(c1), _dev(d2), _dev(c2), _dev(rel), K, factor)
    assert np.array_equal(_u32(counts), want_c.astype(np.int64))
    np.testing.assert_allclose(sums.cpu().numpy(), want_s, rtol=2e-5, atol=1e-6)
