// Overlap of a frame pair straight from the anchor's depth plane, for gfx950 (MI355X): the chain
// _point_cloud -> _transform_pointcloud -> _backtransform_pointcloud -> _overlap of the reference's frame
// dropping (slam/preprocess/test_import.py:148-176, called at :204-214) in one pass that reads 2 bytes per
// pixel and writes two counters per pair, nothing materialised.
//
// Arithmetic contract: the two counts (points kept by _point_cloud, and of those the ones whose projection
// lies in [0, dim_u] x [0, dim_v]) equal what the reference's float64 operation order gives, bit for bit.
// As in the fused pair-loss kernel (pair_loss.hip) a cheap float64 evaluation of the projection decides
// every pixel whose coordinates are farther from the window's four borders than a rigorous bound on the
// distance between the cheap and the reference value; the few pixels within the bound of a border, behind
// the camera or non-finite are redone in the reference's exact order (geom_device.h).  Only the four
// borders matter here (no tap choice), so all but a few pixels in 10^6 are decided by the cheap form.
//
// Most pixels are not projected at all.  For a fixed depth the projection is a projective map of the pixel
// coordinates with a positive denominator (z' > 0), which takes a wave's 128 x RPW rectangle to a convex
// quadrilateral, and as the depth of one pixel varies its projection moves along a segment; so every pixel of
// the rectangle with a depth in [z_min, z_max] projects into the convex hull of the rectangle's four corners
// taken at z_min and at z_max.  When these eight points lie inside the window by the certificate's margin
// (or all beyond one of its borders by that margin), so does every pixel, in the reference's arithmetic too,
// and the wave only counts its non-zero depths: 2.5 instead of 21 vector instructions per pixel.  Waves whose
// rectangle straddles a border project pixel by pixel.
//
// HBM-streaming by its bytes, bound by vector instructions; nothing GEMM-shaped, no MFMA.  Built with
// -ffp-contract=off like the other geometry files.
#include "geom_device.h"

namespace slamhip {

constexpr int OVL_TILE_W = 128;   // a wave's 64 lanes hold two columns each
#ifndef SLAMHIP_OVL_RPW
#define SLAMHIP_OVL_RPW 15
#endif
constexpr int OVL_RPW = SLAMHIP_OVL_RPW;   // rows a lane walks; the tile's four waves lie one below the other

struct OvlArgs {
  Cam cam;
  // host-rounded quotients of the cheap projection (any rounding is covered by the bound)
  double fx_fy, fy_fx, inv_fac, inv_fx_fac, inv_fy_fac;
  double half_u, half_v;   // dim_u / 2, dim_v / 2
  float axm, aym;          // max |col - cx| / fx and max |row - cy| / fy over the image
  float extent;            // bound of |u - cx|, |v - cy| for pixels near the window
  int B, H, W;
  int cert_ok;             // the constants are mid-range: the cheap form may decide pixels
  int cx_col, cy_row;      // the column / row with col - cx == 0 / row - cy == 0, or -1
  int pair_loads;          // W is even and the plane 4-byte aligned: a lane reads its two columns in one load
  int count_bits;          // with a workspace: bits per count in a frame's 64-bit record {kept, inside, tickets}
};

// Constants of the cheap projection that depend on the pair alone (wave-uniform).
struct OvlConsts {
  double kAc, kAr, kA0, kBc, kBr, kB0, kCc, kCr, kC0, t0p, t1p, t2, dz;
  double cu, cv;                       // cx - dim_u / 2, cy - dim_v / 2: the projection comes out centred on the window
  double in_u, out_u, in_v, out_v;     // |centred coordinate| below in_*: inside for certain; above out_*: outside for certain
  int cert;
};

// unsigned min / max over the wave by DPP steps, valid in lane 63 (masked-out rows receive the identity)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ unsigned int ovl_dpp(unsigned int old, unsigned int v) {
  return (unsigned int)__builtin_amdgcn_update_dpp((int)old, (int)v, CTRL, ROW_MASK, 0xF, false);
}
__device__ __forceinline__ unsigned int wave_umin63(unsigned int v) {
  v = min(v, ovl_dpp<0xB1, 0xF>(0xffffffffu, v));    // quad_perm [1,0,3,2]
  v = min(v, ovl_dpp<0x4E, 0xF>(0xffffffffu, v));    // quad_perm [2,3,0,1]
  v = min(v, ovl_dpp<0x141, 0xF>(0xffffffffu, v));   // row_half_mirror
  v = min(v, ovl_dpp<0x140, 0xF>(0xffffffffu, v));   // row_mirror
  v = min(v, ovl_dpp<0x142, 0xA>(0xffffffffu, v));   // row_bcast15 into rows 1, 3
  v = min(v, ovl_dpp<0x143, 0xC>(0xffffffffu, v));   // row_bcast31 into rows 2, 3
  return v;
}
__device__ __forceinline__ unsigned int wave_umax63(unsigned int v) {
  v = max(v, ovl_dpp<0xB1, 0xF>(0u, v));
  v = max(v, ovl_dpp<0x4E, 0xF>(0u, v));
  v = max(v, ovl_dpp<0x141, 0xF>(0u, v));
  v = max(v, ovl_dpp<0x140, 0xF>(0u, v));
  v = max(v, ovl_dpp<0x142, 0xA>(0u, v));
  v = max(v, ovl_dpp<0x143, 0xC>(0u, v));
  return v;
}

typedef unsigned short ovl_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ ovl_u16x2 as_u16x2(unsigned int v) { return __builtin_bit_cast(ovl_u16x2, v); }
__device__ __forceinline__ unsigned int as_u32(ovl_u16x2 v) { return __builtin_bit_cast(unsigned int, v); }
__device__ __forceinline__ unsigned int pk_umin(unsigned int a, unsigned int b) {
  return as_u32(__builtin_elementwise_min(as_u16x2(a), as_u16x2(b)));
}
__device__ __forceinline__ unsigned int pk_umax(unsigned int a, unsigned int b) {
  return as_u32(__builtin_elementwise_max(as_u16x2(a), as_u16x2(b)));
}
__device__ __forceinline__ unsigned int pk_usub(unsigned int a, unsigned int b) { return as_u32(as_u16x2(a) - as_u16x2(b)); }
__device__ __forceinline__ unsigned int pk_uadd(unsigned int a, unsigned int b) { return as_u32(as_u16x2(a) + as_u16x2(b)); }
// sum over the wave, valid in lane 63
__device__ __forceinline__ unsigned int wave_usum63(unsigned int v) {
  v += ovl_dpp<0xB1, 0xF>(0u, v);
  v += ovl_dpp<0x4E, 0xF>(0u, v);
  v += ovl_dpp<0x141, 0xF>(0u, v);
  v += ovl_dpp<0x140, 0xF>(0u, v);
  v += ovl_dpp<0x142, 0xA>(0u, v);
  v += ovl_dpp<0x143, 0xC>(0u, v);
  return v;
}

#ifdef SLAMHIP_AUDIT
// Diagnostic build only (-DSLAMHIP_AUDIT, tools/audit_certificate.py): every pixel that the cheap projection decides
// (a whole rectangle at once, or pixel by pixel) is decided a second time with the reference-order functions of
// geom_device.h.   [0] pixels inside the image   [1] decided by the cheap form   [2] left to the exact form
//                  [3] "inside the window" differs (pixels)                      [6] "kept by _point_cloud" differs
__device__ unsigned long long g_audit_ovl[8];
__device__ __forceinline__ void audit_ovl_add(int i, unsigned long long n) {
  if (n != 0ull && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u) atomicAdd(&g_audit_ovl[i], n);
}
#endif

// A workgroup is a tile of OVL_TILE_W columns x 4 OVL_RPW rows, a wave a rectangle of OVL_TILE_W x OVL_RPW, a lane two
// neighbouring image columns: a row of the lane is one 32-bit register {depth of column c | depth of column c + 1 << 16},
// read with one 4-byte load where the plane allows it, and the statistics of the whole-rectangle decision run on the
// packed pairs.
__global__ void __launch_bounds__(256)
overlap_fused_kernel(const OvlArgs a, const uint16_t* __restrict__ depth, const double* __restrict__ rel,
                     unsigned int* __restrict__ counts, unsigned int* __restrict__ acc) {
  constexpr int RPW = OVL_RPW;
  __shared__ OvlConsts s_c;
  __shared__ unsigned int s_cnt[4][2];

  const int H = a.H, W = a.W;
  const int b = blockIdx.z;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wc0 = blockIdx.x * OVL_TILE_W;        // first column of the rectangle
  const int col = wc0 + 2 * lane;                 // this lane's columns: col, col + 1
  const int row0 = (blockIdx.y * 4 + wave) * RPW;
  const Pose p = load_pose(rel + 16 * (long long)b);

  // this lane's pixels: raw depth, read once; nothing waits for the loads until they are used.  Pixels outside
  // the image count as depth 0 (only the tiles of the right and bottom edge have any).
  const uint16_t* d1b = depth + (long long)b * H * W;
  unsigned int a_d[RPW];
  if (a.pair_loads) {   // W even and the plane 4-byte aligned: a lane's two columns are one aligned 32-bit word
    const unsigned int cc = (unsigned int)min(col, W - 2);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const unsigned int rw = (unsigned int)min(row0 + r, H - 1) * (unsigned int)W;   // (an image has at most 2^29 pixels)
      a_d[r] = *reinterpret_cast<const unsigned int*>(d1b + (size_t)(rw + cc));
    }
    if (col >= W) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) a_d[r] = 0u;
    }
  } else {
    const unsigned int c0 = (unsigned int)min(col, W - 1), c1 = (unsigned int)min(col + 1, W - 1);
    const unsigned int keep = (col < W ? 0xffffu : 0u) | (col + 1 < W ? 0xffff0000u : 0u);
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      const unsigned int rw = (unsigned int)min(row0 + r, H - 1) * (unsigned int)W;
      a_d[r] = ((unsigned int)d1b[(size_t)(rw + c0)] | ((unsigned int)d1b[(size_t)(rw + c1)] << 16)) & keep;
    }
  }
  if (row0 + RPW > H) {
#pragma unroll
    for (int r = 0; r < RPW; ++r) a_d[r] = (row0 + r < H) ? a_d[r] : 0u;
  }
  // points with x == 0 or y == 0 are dropped by _point_cloud: with an integer principal point the pixels of that
  // column and of that row count as depth 0 as well
  if (a.cx_col >= 0 || a.cy_row >= 0) {
    const unsigned int e = (unsigned int)(a.cx_col - col);      // 0 / 1: this lane holds the column
    const unsigned int keep = e == 0u ? 0xffff0000u : e == 1u ? 0x0000ffffu : 0xffffffffu;
#pragma unroll
    for (int r = 0; r < RPW; ++r) a_d[r] = (row0 + r == a.cy_row) ? 0u : (a_d[r] & keep);
  }

  // ---- certificate and constants (wave 0, while the others' loads are in flight) ----
  // With S = 65535/factor * (axm |r0j| + aym |r1j| + |r2j|) + |tj| (the largest sum of magnitudes in the
  // transform of coordinate j), the reference's x', y', z' and the cheap ones each differ from the real-number
  // value by at most 8 resp. 24 roundings of 2^-53 S.  For z' above delta_z = 2^-16 Sz the projected coordinate
  // therefore moves by at most 2^-31 (fx Sx / Sz + |u - cx|), which is below
  //   eps = 2^-30 ((fx Sx + fy Sy) / Sz + extent)
  // for every pixel within twice `extent` of the principal point; farther out the error (2^-31 |u - cx| more) is
  // a vanishing fraction of the distance to the window, which lies within `extent`.  eps_w = max(2 eps, 2^-20)
  // is the margin kept from the window's borders on either side.
  if (wave == 0) {
    const Cam cam = a.cam;
    OvlConsts c;
    const float zmaxf = 65535.0f * (float)a.inv_fac;
    const float Sx = zmaxf * (a.axm * fabsf((float)p.r00) + a.aym * fabsf((float)p.r10) + fabsf((float)p.r20)) + fabsf((float)p.t0);
    const float Sy = zmaxf * (a.axm * fabsf((float)p.r01) + a.aym * fabsf((float)p.r11) + fabsf((float)p.r21)) + fabsf((float)p.t1);
    const float Sz = zmaxf * (a.axm * fabsf((float)p.r02) + a.aym * fabsf((float)p.r12) + fabsf((float)p.r22)) + fabsf((float)p.t2);
    const float delta_z = Sz * 0x1p-16f;
    const float ratio = ((float)cam.fx * Sx + (float)cam.fy * Sy) / Sz;
    const float eps = 0x1p-30f * (ratio + a.extent);
    const float eps_w = fmaxf(2.0f * eps, 0x1p-20f);
    c.in_u = a.half_u - (double)eps_w; c.out_u = a.half_u + (double)eps_w;
    c.in_v = a.half_v - (double)eps_w; c.out_v = a.half_v + (double)eps_w;
    c.cert = (a.cert_ok != 0) && (eps_w <= 0x1p-7f) && (delta_z > 0.0f) && (delta_z < 3.0e38f) &&
             (c.in_u > 0.0) && (c.in_v > 0.0);   // NaN fails
    c.dz = (double)delta_z;
    // u - dim_u/2 = (z A + t0') / (z C + t2) + (cx - dim_u/2) with
    //   A = (col-cx) r00 + (row-cy) (fx/fy) r10 + fx r20,  C = ((col-cx)/fx) r02 + ((row-cy)/fy) r12 + r22
    // (all divided by factor: z is the raw depth), likewise v; A, B, C advance by a constant per row
    c.kAc = p.r00 * a.inv_fac; c.kAr = (a.fx_fy * p.r10) * a.inv_fac; c.kA0 = (cam.fx * p.r20) * a.inv_fac;
    c.kBc = (a.fy_fx * p.r01) * a.inv_fac; c.kBr = p.r11 * a.inv_fac; c.kB0 = (cam.fy * p.r21) * a.inv_fac;
    c.kCc = p.r02 * a.inv_fx_fac; c.kCr = p.r12 * a.inv_fy_fac; c.kC0 = p.r22 * a.inv_fac;
    c.t0p = cam.fx * p.t0; c.t1p = cam.fy * p.t1; c.t2 = p.t2;
    c.cu = cam.cx - a.half_u; c.cv = cam.cy - a.half_v;
    if (lane == 0) s_c = c;
  }
  __syncthreads();
  const OvlConsts pc = s_c;
  const bool cert_u = __builtin_amdgcn_readfirstlane(pc.cert) != 0;

  unsigned int n_valid0 = 0u, n_inb = 0u;     // wave-uniform counts of the decided pixels
  unsigned int rest0 = 0u, rest1 = 0u;        // bit r: row r of this lane's first / second column is left to the exact form

  // ---- the wave's rectangle as a whole: depth range, number of kept points, the eight hull points ----
  bool whole = false;
  if (cert_u) {
    unsigned int mn = 0xffffffffu, mx = 0u, cnt = 0u;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
      mn = pk_umin(mn, pk_usub(a_d[r], 0x00010001u));   // (depth 0 wraps to the identity of the minimum)
      mx = pk_umax(mx, a_d[r]);
      cnt = pk_uadd(cnt, pk_umin(a_d[r], 0x00010001u));
    }
    const unsigned int n_kept = (unsigned int)__builtin_amdgcn_readlane((int)wave_usum63((cnt & 0xffffu) + (cnt >> 16)), 63);
    const unsigned int z_lo = ((unsigned int)__builtin_amdgcn_readlane((int)wave_umin63(min(mn & 0xffffu, mn >> 16)), 63) + 1u) & 0xffffu;
    const unsigned int z_hi = (unsigned int)__builtin_amdgcn_readlane((int)wave_umax63(max(mx & 0xffffu, mx >> 16)), 63);
    if (n_kept == 0u) {
      whole = true;   // no depth at all: nothing kept
    } else {
      // lane l < 8: corner (l & 1 ? last : first column, l & 2 ? last : first row) at (l & 4 ? z_hi : z_lo)
      const double ccol = (double)((lane & 1) ? min(wc0 + OVL_TILE_W - 1, W - 1) : wc0) - a.cam.cx;
      const double crow = (double)((lane & 2) ? min(row0 + RPW - 1, H - 1) : row0) - a.cam.cy;
      const double z = (double)((lane & 4) ? z_hi : z_lo);
      const double Ac = fma(ccol, pc.kAc, fma(crow, pc.kAr, pc.kA0));
      const double Bc = fma(ccol, pc.kBc, fma(crow, pc.kBr, pc.kB0));
      const double Cc = fma(ccol, pc.kCc, fma(crow, pc.kCr, pc.kC0));
      const double X = fma(z, Ac, pc.t0p), Y = fma(z, Bc, pc.t1p), Z = fma(z, Cc, pc.t2);
      double rcp = __builtin_amdgcn_rcp(Z);
      rcp = fma(rcp, fma(-Z, rcp, 1.0), rcp);
      const double su = fma(X, rcp, pc.cu), sv = fma(Y, rcp, pc.cv);   // centred on the window
      auto all8 = [&](bool c) { return (__builtin_amdgcn_ballot_w64(c) & 0xffull) == 0xffull; };   // NaN fails
      if (all8(Z > pc.dz)) {
        if (all8(fabs(su) < pc.in_u) && all8(fabs(sv) < pc.in_v)) {
          whole = true;
          n_valid0 = n_kept; n_inb = n_kept;
        } else if (all8(su < -pc.out_u) || all8(su > pc.out_u) || all8(sv < -pc.out_v) || all8(sv > pc.out_v)) {
          whole = true;
          n_valid0 = n_kept;
        }
      }
    }
  }

#ifdef SLAMHIP_AUDIT
  // the exact decision of one pixel (raw depth as masked above: pixels outside the image and the principal
  // point's row / column hold 0)
  auto audit_exact = [&](int row, int cole, unsigned int draw, bool& v0, bool& inb) {
    double x, y, z, x2, y2, z2, u, v;
    backproject_px(a.cam, cole, row, draw, x, y, z);
    v0 = (x != 0.0) & (y != 0.0) & (z != 0.0);
    transform_px(p, x, y, z, x2, y2, z2);
    project_px(a.cam, x2, y2, z2, u, v);
    inb = v0 && in_window(u, v, 2.0 * a.half_u, 2.0 * a.half_v);
  };
  {
    unsigned long long n_img = 0, e_valid0 = 0, e_inb = 0, n_v0diff = 0;
    for (int r = 0; r < RPW; ++r)
      for (int e = 0; e < 2; ++e) {
        const bool inside = row0 + r < H && col + e < W;
        const unsigned int draw = e ? (a_d[r] >> 16) : (a_d[r] & 0xffffu);
        const unsigned int raw = inside ? d1b[(long long)(row0 + r) * W + col + e] : 0u;   // unmasked
        bool v0 = false, inb = false;
        if (inside) audit_exact(row0 + r, col + e, raw, v0, inb);
        n_img += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(inside));
        e_valid0 += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(v0));
        e_inb += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(inb));
        n_v0diff += (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(inside && (v0 != (draw != 0u))));
      }
    audit_ovl_add(0, n_img);
    audit_ovl_add(6, n_v0diff);
    if (whole) {
      audit_ovl_add(1, n_img);
      audit_ovl_add(3, (e_valid0 > n_valid0 ? e_valid0 - n_valid0 : n_valid0 - e_valid0) +
                           (e_inb > n_inb ? e_inb - n_inb : n_inb - e_inb));
    }
  }
#endif
  if (whole) {
    // decided above
  } else if (cert_u) {
    // ---- pixel by pixel: the cheap projection decides all but the pixels near a border ----
    const double ucx = (double)col - a.cam.cx, vcy0 = (double)row0 - a.cam.cy;
    double A = fma(ucx, pc.kAc, fma(vcy0, pc.kAr, pc.kA0));
    double Bq = fma(ucx, pc.kBc, fma(vcy0, pc.kBr, pc.kB0));
    double C = fma(ucx, pc.kCc, fma(vcy0, pc.kCr, pc.kC0));
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const unsigned int draw = e ? (a_d[r] >> 16) : (a_d[r] & 0xffffu);
        const double Ae = e ? A + pc.kAc : A, Be = e ? Bq + pc.kBc : Bq, Ce = e ? C + pc.kCc : C;
        const double z = (double)draw;
        const double X = fma(z, Ae, pc.t0p), Y = fma(z, Be, pc.t1p), Z = fma(z, Ce, pc.t2);
        double rcp = __builtin_amdgcn_rcp(Z);
        rcp = fma(rcp, fma(-Z, rcp, 1.0), rcp);
        const double du = fabs(fma(X, rcp, pc.cu)), dv = fabs(fma(Y, rcp, pc.cv));   // distance from the window's centre
        const unsigned long long m_valid0 = __builtin_amdgcn_ballot_w64(draw != 0u);
        const unsigned long long m_z = m_valid0 & __builtin_amdgcn_ballot_w64(Z > pc.dz);
        const unsigned long long m_in = m_z & __builtin_amdgcn_ballot_w64(du < pc.in_u) & __builtin_amdgcn_ballot_w64(dv < pc.in_v);
        const unsigned long long m_out = m_z & (__builtin_amdgcn_ballot_w64(du > pc.out_u) | __builtin_amdgcn_ballot_w64(dv > pc.out_v));
        n_valid0 += (unsigned int)__popcll(m_in | m_out);
        n_inb += (unsigned int)__popcll(m_in);
#ifdef SLAMHIP_AUDIT
        {
          bool v0 = false, inb = false;
          if (row0 + r < H && col + e < W) audit_exact(row0 + r, col + e, draw, v0, inb);
          const unsigned long long m_e = __builtin_amdgcn_ballot_w64(inb);
          audit_ovl_add(1, (unsigned long long)__popcll(m_in | m_out));
          audit_ovl_add(2, (unsigned long long)__popcll(m_valid0 & ~(m_in | m_out)));
          audit_ovl_add(3, (unsigned long long)__popcll((m_in & ~m_e) | (m_out & m_e)));
        }
#endif
        const unsigned long long m_rest = m_valid0 & ~(m_in | m_out);   // near a border, behind the camera, non-finite
        if (m_rest != 0ull) {
          const unsigned int bit = __builtin_amdgcn_inverse_ballot_w64(m_rest) ? (1u << r) : 0u;
          if (e) rest1 |= bit; else rest0 |= bit;
        }
      }
      A += pc.kAr; Bq += pc.kBr; C += pc.kCr;
    }
  } else {
    // no certificate (constants out of range, a degenerate pose): every pixel in the image
#ifdef SLAMHIP_AUDIT
    for (int r = 0; r < RPW; ++r)
      for (int e = 0; e < 2; ++e)
        audit_ovl_add(2, (unsigned long long)__popcll(__builtin_amdgcn_ballot_w64(row0 + r < H && col + e < W)));
#endif
    const int n_in = min(max(H - row0, 0), RPW);
    const unsigned int rows_in = (1u << n_in) - 1u;     // RPW < 32
    rest0 = col < W ? rows_in : 0u;
    rest1 = col + 1 < W ? rows_in : 0u;
  }

  // ---- the reference's exact operation order for the pixels left ----
  if (__builtin_amdgcn_ballot_w64((rest0 | rest1) != 0u) != 0ull) {
    const double dim_u = 2.0 * a.half_u, dim_v = 2.0 * a.half_v;   // (exact: the halves are the host's dim / 2)
#pragma unroll 1
    for (int r = 0; r < RPW; ++r) {
#pragma unroll 1
      for (int e = 0; e < 2; ++e) {
        const bool need = ((e ? rest1 : rest0) >> r) & 1u;
        if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
        bool valid0 = false, inb = false;
        if (need) {
          const int row = row0 + r, cole = col + e;
          const unsigned int draw = d1b[(long long)row * W + cole];
          double x, y, z, x2, y2, z2, u, v;
          backproject_px(a.cam, cole, row, draw, x, y, z);
          valid0 = (x != 0.0) & (y != 0.0) & (z != 0.0);
          transform_px(p, x, y, z, x2, y2, z2);
          project_px(a.cam, x2, y2, z2, u, v);
          inb = valid0 && in_window(u, v, dim_u, dim_v);
        }
        n_valid0 += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(valid0));
        n_inb += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(inb));
      }
    }
  }

  // ---- the four waves through LDS, one atomic instruction per workgroup ----
  if (lane == 0) { s_cnt[wave][0] = n_valid0; s_cnt[wave][1] = n_inb; }
  __syncthreads();
  if (acc && a.count_bits > 0) {
    // Workspace record of the frame as ONE 64-bit word {kept | inside << n | tickets << 2 n}: a workgroup adds its
    // two counts and its ticket with a single atomic instruction and learns from the value that comes back whether
    // it was the last of the frame - then old + its own are the totals, which it moves to `counts`, and it leaves
    // the record zero for the next call.  One round trip to the L2 per workgroup, no clearing kernel.
    if (tid == 0) {
      const unsigned int n = (unsigned int)a.count_bits;
      const unsigned long long c0 = s_cnt[0][0] + s_cnt[1][0] + s_cnt[2][0] + s_cnt[3][0];
      const unsigned long long c1 = s_cnt[0][1] + s_cnt[1][1] + s_cnt[2][1] + s_cnt[3][1];
      const unsigned long long mine = c0 | (c1 << n) | (1ull << (2u * n));
      unsigned long long* rec = reinterpret_cast<unsigned long long*>(acc + 4 * (long long)b);
      const unsigned long long old = atomicAdd(rec, mine);
      if ((old >> (2u * n)) == (unsigned long long)(gridDim.x * gridDim.y - 1u)) {
        const unsigned long long tot = old + mine, mask = (1ull << n) - 1ull;
        counts[2 * (long long)b] = (unsigned int)(tot & mask);
        counts[2 * (long long)b + 1] = (unsigned int)((tot >> n) & mask);
        atomicExch(rec, 0ull);
      }
    }
    return;
  }
  if (tid < 2) {
    const unsigned int c = s_cnt[0][tid] + s_cnt[1][tid] + s_cnt[2][tid] + s_cnt[3][tid];
    if (c) atomicAdd(&counts[2 * (long long)b + tid], c);
  }
}

}  // namespace slamhip

using namespace slamhip;

static int launch_overlap_fused(const uint16_t* depth, const double* rel4x4, const double K[4], double factor,
                                double dim_u, double dim_v, int B, int H, int W, uint32_t* counts, unsigned int* acc,
                                void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, SLAMHIP_E_ARG, "overlap_fused: bad shape B=%d H=%d W=%d", B, H, W);
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(depth && rel4x4 && K && counts, SLAMHIP_E_ARG, "overlap_fused: null pointer");
  SLAMHIP_REQUIRE(H <= 32767 && W <= 65535 && (long long)H * W <= 0x20000000LL, SLAMHIP_E_SIZE,
                  "overlap_fused: image larger than 65535 x 32767 or 2^29 pixels");
  SLAMHIP_REQUIRE((reinterpret_cast<uintptr_t>(depth) % 2) == 0 && (reinterpret_cast<uintptr_t>(rel4x4) % 8) == 0 &&
                      (reinterpret_cast<uintptr_t>(counts) % 4) == 0,
                  SLAMHIP_E_ARG, "overlap_fused: misaligned input or output");
  const int tiles_x = (W + OVL_TILE_W - 1) / OVL_TILE_W, tiles_y = (H + 4 * OVL_RPW - 1) / (4 * OVL_RPW);
  SLAMHIP_REQUIRE(B <= 65535 && tiles_y <= 65535, SLAMHIP_E_SIZE, "overlap_fused: more than 65535 pairs per call");
  hipStream_t s = (hipStream_t)stream;
  OvlArgs a;
  a.cam = Cam{K[0], K[1], K[2], K[3], factor};
  a.B = B; a.H = H; a.W = W;
  // the cheap form assumes that no intermediate comes near the denormal or overflow range ("x, y, z all non-zero"
  // is then "raw depth, col - cx and row - cy all non-zero"): constants in [2^-64, 2^64], principal point in
  // [2^-32, 2^32], a finite window; its error bound covers the roundings of the quotients below
  auto mid = [](double v) { const double m = fabs(v); return m >= 0x1p-64 && m <= 0x1p64; };
  auto pp = [](double v) { const double m = fabs(v); return m >= 0x1p-32 && m <= 0x1p32; };
  a.cert_ok = mid(K[0]) && mid(K[1]) && mid(factor) && pp(K[2]) && pp(K[3]) && dim_u > 0.0 && dim_v > 0.0 &&
              dim_u <= 0x1p24 && dim_v <= 0x1p24;
  a.fx_fy = K[0] / K[1]; a.fy_fx = K[1] / K[0];
  a.inv_fac = 1.0 / factor; a.inv_fx_fac = 1.0 / (K[0] * factor); a.inv_fy_fac = 1.0 / (K[1] * factor);
  a.half_u = 0.5 * dim_u; a.half_v = 0.5 * dim_v;
  a.axm = (float)(fmax(fabs(K[2]), fabs((double)(W - 1) - K[2])) / fabs(K[0]) * 1.0001);
  a.aym = (float)(fmax(fabs(K[3]), fabs((double)(H - 1) - K[3])) / fabs(K[1]) * 1.0001);
  // bound of |u - cx|, |v - cy| over the window and the image together
  a.extent = (float)(fmax((double)W, dim_u) + fmax((double)H, dim_v) + fabs(K[2]) + fabs(K[3]) + 8.0);
  auto int_index = [](double c, int n) { return (c >= 0.0 && c < (double)n && c == floor(c)) ? (int)c : -1; };
  a.cx_col = int_index(K[2], W);
  a.cy_row = int_index(K[3], H);
  a.pair_loads = (W % 2 == 0) && W >= 2 && (reinterpret_cast<uintptr_t>(depth) % 4) == 0;
  // bits for a count of up to H W pixels and for tiles_x tiles_y tickets: packed into one word when they fit
  {
    auto bits = [](unsigned long long v) { int n = 0; while (v) { ++n; v >>= 1; } return n; };
    const int nb = bits((unsigned long long)H * W), tb = bits((unsigned long long)tiles_x * tiles_y);
    a.count_bits = acc ? nb : 0;
    SLAMHIP_REQUIRE(!acc || 2 * nb + tb <= 64, SLAMHIP_E_SIZE,
                    "overlap_fused_ws: %d x %d frames need %d bits per workspace record (64 available); use slamhip_overlap_fused",
                    W, H, 2 * nb + tb);
  }
  // without a workspace the counters are cleared by a kernel on the stream (a call recorded in a HIP graph holds
  // kernel nodes only); with one the call is this launch alone
  if (!acc) clear_u32_async(counts, 2LL * B, s);
  hipLaunchKernelGGL(overlap_fused_kernel, dim3((unsigned)tiles_x, (unsigned)tiles_y, (unsigned)B), dim3(256), 0, s, a,
                     depth, rel4x4, counts, acc);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

extern "C" int slamhip_overlap_fused(const uint16_t* depth, const double* rel4x4, const double K[4], double factor,
                                     double dim_u, double dim_v, int B, int H, int W, uint32_t* counts, void* stream) {
  return launch_overlap_fused(depth, rel4x4, K, factor, dim_u, dim_v, B, H, W, counts, nullptr, stream);
}

extern "C" size_t slamhip_overlap_fused_workspace_bytes(int B) { return B > 0 ? 16 * (size_t)B : 0; }

extern "C" int slamhip_overlap_fused_ws(const uint16_t* depth, const double* rel4x4, const double K[4], double factor,
                                        double dim_u, double dim_v, int B, int H, int W, uint32_t* counts,
                                        void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(B <= 0 || (workspace && workspace_bytes >= slamhip_overlap_fused_workspace_bytes(B) &&
                             (reinterpret_cast<uintptr_t>(workspace) % 16) == 0),
                  SLAMHIP_E_ARG, "overlap_fused_ws: workspace of %zu bytes (need %zu, 16-byte aligned)", workspace_bytes,
                  slamhip_overlap_fused_workspace_bytes(B));
  return launch_overlap_fused(depth, rel4x4, K, factor, dim_u, dim_v, B, H, W, counts, (unsigned int*)workspace, stream);
}

#ifdef SLAMHIP_AUDIT
// audit build only: copies the eight counters to the host (after draining the device) and optionally clears them
extern "C" int slamhip_audit_ovl_read(unsigned long long out[8], int reset) {
  SLAMHIP_HIP(hipDeviceSynchronize());
  SLAMHIP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_audit_ovl), 8 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SLAMHIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_audit_ovl), z, sizeof(z)));
  }
  return SLAMHIP_OK;
}
#endif
