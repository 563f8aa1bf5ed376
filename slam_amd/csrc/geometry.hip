// Per-pixel frame-pair geometry for gfx950 (MI355X): back-projection, rigid
// transform, re-projection, in-image predicate, bilinear warp of the neighbour
// frame and the residual / mask reductions.
//
// Arithmetic contract (follows slam/preprocess/test_import.py:148-176): every
// coordinate is IEEE float64 evaluated in the reference's operation order; this
// file is built with -ffp-contract=off so no product/sum pair is fused except
// where fma() is written (the reference's `np.dot(points, R)` is a dgemm that
// accumulates k = 0,1,2 with fused multiply-adds).  Masks, counts and pixel
// indices therefore equal the reference's numpy results bit for bit.  Colour
// and depth interpolation runs in float32.
//
// The kernels are HBM-bound streaming kernels with a float64 VALU load; nothing
// here is GEMM-shaped, so no MFMA.
#include <hip/hip_runtime.h>

#include <atomic>
#include <climits>
#include <cmath>
#include <cstdint>

#include "common.h"

namespace slamhip {

struct Cam {
  double fx, fy, cx, cy, factor;
};

struct Pose {  // rel[0:3,0:3] and rel[0:3,3] of a row-major 4x4
  double r00, r01, r02, r10, r11, r12, r20, r21, r22, t0, t1, t2;
};

__device__ __forceinline__ Pose load_pose(const double* __restrict__ m) {
  Pose p;
  p.r00 = m[0]; p.r01 = m[1]; p.r02 = m[2]; p.t0 = m[3];
  p.r10 = m[4]; p.r11 = m[5]; p.r12 = m[6]; p.t1 = m[7];
  p.r20 = m[8]; p.r21 = m[9]; p.r22 = m[10]; p.t2 = m[11];
  return p;
}

// test_import.py:153-155: z = d/FACTOR; x = ((u-cx)*z)/fx; y = ((v-cy)*z)/fy
__device__ __forceinline__ void backproject_px(const Cam& c, int u, int v, uint32_t raw, double& x,
                                               double& y, double& z) {
  z = (double)raw / c.factor;
  x = (((double)u - c.cx) * z) / c.fx;
  y = (((double)v - c.cy) * z) / c.fy;
}

// test_import.py:160: [x y z] . R + t, dgemm order (k ascending, fused accumulate)
__device__ __forceinline__ void transform_px(const Pose& p, double x, double y, double z, double& x2,
                                             double& y2, double& z2) {
  x2 = fma(z, p.r20, fma(y, p.r10, x * p.r00)) + p.t0;
  y2 = fma(z, p.r21, fma(y, p.r11, x * p.r01)) + p.t1;
  z2 = fma(z, p.r22, fma(y, p.r12, x * p.r02)) + p.t2;
}

// test_import.py:165-167: u = x*fx/z + cx; v = y*fy/z + cy; d = z
__device__ __forceinline__ void project_px(const Cam& c, double x2, double y2, double z2, double& u,
                                           double& v) {
  u = (x2 * c.fx) / z2 + c.cx;
  v = (y2 * c.fy) / z2 + c.cy;
}

// test_import.py:174 (inclusive on both ends, NaN fails)
__device__ __forceinline__ bool in_window(double u, double v, double du, double dv) {
  return (0.0 <= u) & (u <= du) & (0.0 <= v) & (v <= dv);   // '&': four compares, no branches
}

// floor to int32, saturating, NaN -> INT32_MIN
__device__ __forceinline__ int32_t pixel_index(double c) {
  if (c != c) return INT_MIN;
  const double f = floor(c);
  if (f <= -2147483648.0) return INT_MIN;
  if (f >= 2147483647.0) return INT_MAX;
  return (int32_t)f;
}

template <typename T>
__device__ __forceinline__ void store3(T* p, double a, double b, double c) {
  p[0] = (T)a;
  p[1] = (T)b;
  p[2] = (T)c;
}

// ---------------------------------------------------------------------------
// materialising kernels (parity / facade use; one pixel per thread)
// ---------------------------------------------------------------------------
template <typename OutT>
__global__ void __launch_bounds__(256)
backproject_kernel(const uint16_t* __restrict__ depth, Cam cam, int H, int W, long long total,
                   OutT* __restrict__ xyz, uint8_t* __restrict__ valid0) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int u = (int)(i % W);
  const int v = (int)((i / W) % H);
  double x, y, z;
  backproject_px(cam, u, v, depth[i], x, y, z);
  if (xyz) store3(xyz + 3 * i, x, y, z);
  if (valid0) valid0[i] = (x != 0.0 && y != 0.0 && z != 0.0) ? 1 : 0;
}

template <typename InT, typename OutT>
__global__ void __launch_bounds__(256)
reproject_kernel(const InT* __restrict__ xyz, const uint8_t* __restrict__ valid0,
                 const double* __restrict__ rel, Cam cam, double dim_u, double dim_v, int HW,
                 int blocks_per_image, OutT* __restrict__ xyz2, OutT* __restrict__ uvd,
                 int32_t* __restrict__ iu_iv, uint8_t* __restrict__ inb,
                 unsigned int* __restrict__ counts) {
  // a workgroup never straddles two images, so the pose is wave-uniform and the
  // counts need one atomic per wave
  const int b = blockIdx.x / blocks_per_image;
  const int px = (blockIdx.x - b * blocks_per_image) * 256 + threadIdx.x;
  bool v0 = false, ib = false;
  if (px < HW) {
    const long long i = (long long)b * HW + px;
    const double x = (double)xyz[3 * i], y = (double)xyz[3 * i + 1], z = (double)xyz[3 * i + 2];
    v0 = valid0 ? (valid0[i] != 0) : (x != 0.0 && y != 0.0 && z != 0.0);
    double x2 = x, y2 = y, z2 = z, u, v;
    if (rel) transform_px(load_pose(rel + 16 * (long long)b), x, y, z, x2, y2, z2);
    project_px(cam, x2, y2, z2, u, v);
    ib = v0 && in_window(u, v, dim_u, dim_v);
    if (xyz2) store3(xyz2 + 3 * i, x2, y2, z2);
    if (uvd) store3(uvd + 3 * i, u, v, z2);
    if (iu_iv) {
      iu_iv[2 * i] = pixel_index(u);
      iu_iv[2 * i + 1] = pixel_index(v);
    }
    if (inb) inb[i] = ib ? 1 : 0;
  }
  if (counts) {
    const unsigned int n0 = (unsigned int)__popcll(__ballot(v0));
    const unsigned int n1 = (unsigned int)__popcll(__ballot(ib));
    if ((threadIdx.x & 63) == 0) {
      if (n0) atomicAdd(&counts[2 * b], n0);
      if (n1) atomicAdd(&counts[2 * b + 1], n1);
    }
  }
}

// test_import.py:171-176 on an already projected [n,3] array
__global__ void __launch_bounds__(256)
overlap_kernel(const double* __restrict__ uvd, long long n, double dim_u, double dim_v,
               uint8_t* __restrict__ mask, unsigned int* __restrict__ count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  bool ib = false;
  if (i < n) {
    ib = in_window(uvd[3 * i], uvd[3 * i + 1], dim_u, dim_v);
    if (mask) mask[i] = ib ? 1 : 0;
  }
  const unsigned int c = (unsigned int)__popcll(__ballot(ib));
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

__global__ void relative_pose_kernel(const float* __restrict__ ga, const float* __restrict__ gb, int n,
                                     double* __restrict__ rel) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double T[2][16];
  for (int s = 0; s < 2; ++s) {
    const float* g = (s == 0 ? ga : gb) + 7 * i;
    const float qx = g[3], qy = g[4], qz = g[5], qw = g[6];
    double* M = T[s];
    M[0] = (double)(((qw * qw + qx * qx) - qy * qy) - qz * qz);
    M[4] = (double)(2.0f * (qx * qy + qw * qz));
    M[8] = (double)(2.0f * (qz * qx - qw * qy));
    M[1] = (double)(2.0f * (qx * qy - qw * qz));
    M[5] = (double)(((qw * qw - qx * qx) + qy * qy) - qz * qz);
    M[9] = (double)(2.0f * (qy * qz + qw * qx));
    M[2] = (double)(2.0f * (qz * qx + qw * qy));
    M[6] = (double)(2.0f * (qy * qz - qw * qx));
    M[10] = (double)(((qw * qw - qx * qx) - qy * qy) + qz * qz);
    M[3] = (double)g[0]; M[7] = (double)g[1]; M[11] = (double)g[2];
    M[12] = M[13] = M[14] = 0.0; M[15] = 1.0;
  }
  double inv[16];
  for (int k = 0; k < 16; ++k) inv[k] = 0.0;
  for (int r = 0; r < 3; ++r)
    for (int c = 0; c < 3; ++c) inv[r * 4 + c] = T[0][c * 4 + r];
  for (int r = 0; r < 3; ++r) {
    double acc = (-inv[r * 4 + 0]) * T[0][3];
    acc = fma(-inv[r * 4 + 1], T[0][7], acc);
    acc = fma(-inv[r * 4 + 2], T[0][11], acc);
    inv[r * 4 + 3] = acc;
  }
  inv[15] = 1.0;
  for (int r = 0; r < 4; ++r)
    for (int c = 0; c < 4; ++c) {
      double acc = inv[r * 4 + 0] * T[1][c];
      for (int k = 1; k < 4; ++k) acc = fma(inv[r * 4 + k], T[1][k * 4 + c], acc);
      rel[16 * (long long)i + r * 4 + c] = acc;
    }
}

// ---------------------------------------------------------------------------
// bilinear taps
// ---------------------------------------------------------------------------
struct Tap {      // one neighbour-frame pixel: colour as three floats, raw depth as float
  float r, g, b, d;
};

__device__ __forceinline__ Tap unpack_cell(uint2 c) {
  Tap t;
  t.r = (float)(c.x & 0xffu);
  t.g = (float)((c.x >> 8) & 0xffu);
  t.b = (float)((c.x >> 16) & 0xffu);
  t.d = __uint_as_float(c.y);
  return t;
}

__device__ __forceinline__ Tap global_tap(const uint16_t* __restrict__ d2, const uint8_t* __restrict__ c2,
                                          int W, int x, int y) {
  const long long o = (long long)y * W + x;
  Tap t;
  t.r = (float)c2[3 * o];
  t.g = (float)c2[3 * o + 1];
  t.b = (float)c2[3 * o + 2];
  t.d = (float)d2[o];
  return t;
}

// top = a + fu*(b-a); bot = c + fu*(e-c); val = top + fv*(bot-top)   (float32, unfused)
__device__ __forceinline__ float lerp2(float a, float b, float c, float e, float fu, float fv) {
  const float top = a + fu * (b - a);
  const float bot = c + fu * (e - c);
  return top + fv * (bot - top);
}

struct WarpOut {
  float wr, wg, wb, wd;   // warped colour, warped depth (already divided by factor)
  bool geo;
};

__device__ __forceinline__ WarpOut blend(const Tap& a, const Tap& b, const Tap& c, const Tap& e, float fu,
                                         float fv, float factor_f) {
  WarpOut o;
  o.wr = lerp2(a.r, b.r, c.r, e.r, fu, fv);
  o.wg = lerp2(a.g, b.g, c.g, e.g, fu, fv);
  o.wb = lerp2(a.b, b.b, c.b, e.b, fu, fv);
  o.wd = lerp2(a.d, b.d, c.d, e.d, fu, fv) / factor_f;
  o.geo = (a.d != 0.0f) && (b.d != 0.0f) && (c.d != 0.0f) && (e.d != 0.0f);
  return o;
}

// ---------------------------------------------------------------------------
// materialising warp kernel (global gathers; the independent cross-check of the
// fused kernel and the source of the facade's per-pixel outputs)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
warp_materialize_kernel(const uint16_t* __restrict__ depth1, const uint8_t* __restrict__ rgb1,
                        const uint16_t* __restrict__ depth2, const uint8_t* __restrict__ rgb2,
                        const double* __restrict__ rel, Cam cam, int k, int H, int W, long long total,
                        float* __restrict__ warped_rgb, float* __restrict__ warped_depth,
                        float* __restrict__ r_photo, float* __restrict__ r_geo,
                        uint8_t* __restrict__ mask, int32_t* __restrict__ iu_iv,
                        double* __restrict__ uvd) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over [B,k,H,W]
  if (i >= total) return;
  const long long HW = (long long)H * W;
  const long long bj = i / HW;          // b*k + j
  const long long b = bj / k;
  const long long px = i % HW;
  const int u0 = (int)(px % W), v0 = (int)(px / W);
  const Pose p = load_pose(rel + 16 * bj);
  double x, y, z, x2, y2, z2, u, v;
  backproject_px(cam, u0, v0, depth1[b * HW + px], x, y, z);
  transform_px(p, x, y, z, x2, y2, z2);
  project_px(cam, x2, y2, z2, u, v);
  const bool valid0 = (x != 0.0 && y != 0.0 && z != 0.0);
  const bool inb = valid0 && in_window(u, v, (double)W, (double)H);
  const bool wv = valid0 && in_window(u, v, (double)(W - 1), (double)(H - 1)) && (z2 > 0.0);
  float w[4] = {0.f, 0.f, 0.f, 0.f}, rp[3] = {0.f, 0.f, 0.f}, rg = 0.f;
  bool geo = false;
  if (wv) {
    const double fl_u = floor(u), fl_v = floor(v);
    const int x0 = (int)fl_u, y0 = (int)fl_v;
    const int x1 = min(x0 + 1, W - 1), y1 = min(y0 + 1, H - 1);
    const float fu = (float)(u - fl_u), fv = (float)(v - fl_v);
    const uint16_t* d2 = depth2 + bj * HW;
    const uint8_t* c2 = rgb2 + 3 * bj * HW;
    const Tap a = global_tap(d2, c2, W, x0, y0), bb = global_tap(d2, c2, W, x1, y0);
    const Tap c = global_tap(d2, c2, W, x0, y1), e = global_tap(d2, c2, W, x1, y1);
    const WarpOut o = blend(a, bb, c, e, fu, fv, (float)cam.factor);
    geo = o.geo;
    w[0] = o.wr; w[1] = o.wg; w[2] = o.wb; w[3] = o.wd;
    const uint8_t* c1 = rgb1 + 3 * (b * HW + px);
    rp[0] = o.wr - (float)c1[0];
    rp[1] = o.wg - (float)c1[1];
    rp[2] = o.wb - (float)c1[2];
    if (geo) rg = (float)((double)o.wd - z2);
  }
  if (warped_rgb) { warped_rgb[3 * i] = w[0]; warped_rgb[3 * i + 1] = w[1]; warped_rgb[3 * i + 2] = w[2]; }
  if (warped_depth) warped_depth[i] = w[3];
  if (r_photo) { r_photo[3 * i] = rp[0]; r_photo[3 * i + 1] = rp[1]; r_photo[3 * i + 2] = rp[2]; }
  if (r_geo) r_geo[i] = rg;
  if (mask)
    mask[i] = (uint8_t)((valid0 ? SLAMHIP_M_VALID0 : 0) | (inb ? SLAMHIP_M_INB : 0) |
                        (wv ? SLAMHIP_M_WARP : 0) | (geo ? SLAMHIP_M_GEO : 0));
  if (iu_iv) { iu_iv[2 * i] = pixel_index(u); iu_iv[2 * i + 1] = pixel_index(v); }
  if (uvd) { uvd[3 * i] = u; uvd[3 * i + 1] = v; uvd[3 * i + 2] = z2; }
}

// ---------------------------------------------------------------------------
// fused frame-pair kernel
// ---------------------------------------------------------------------------
// One 256-thread workgroup owns a TILE_W x TILE_H tile of anchor pixels.  Wave w
// owns rows 4w..4w+3 of the tile and walks them in eight steps of 64 consecutive
// pixels, so that at every step the 64 lanes of a wave read 64 adjacent anchor
// pixels (coalesced) and gather from adjacent LDS cells (no bank conflicts for
// the smooth flow of small camera motions).  Per neighbour frame:
//   A  predict the window of the neighbour frame the tile's taps fall into (the
//      tile's corners projected in float32 at the nearest and farthest depth);
//   B  stage that window into LDS as 8-byte cells {half(1024+R) | half(1024+G),
//      half(1024+B) | raw depth} with 16-byte / 8-byte global loads;
//   C  per pixel: project in float64 (the reference's operation order), gather
//      the four taps from LDS, blend, accumulate; a tap outside the window
//      (large parallax, a window cut at WIN_PITCH x WIN_ROWS) is gathered
//      straight from global memory, so the prediction never affects results.
// Predicates are 64-bit lane masks in scalar registers (one vector compare each,
// combined and counted on the scalar unit).  Reduction: sums by DPP row
// reductions, the four waves through LDS, then one float64 / uint32 atomic per
// quantity per workgroup into the accumulators of the caller's workspace.  The
// workgroup that draws the last ticket of an (image, neighbour) pair moves the
// totals to the outputs with atomic exchanges, which also leaves the workspace
// zeroed for the next call: no memset and no second kernel on the stream.
constexpr int TILE_W = 128;
constexpr int TILE_H = 16;
constexpr int TPB = 256;
#ifndef SLAMHIP_HALVES
#define SLAMHIP_HALVES 1
#endif
constexpr int HALVES = SLAMHIP_HALVES;           // a tile may be walked as two 128 x 8 halves
constexpr int STEPS = (TILE_W * TILE_H) / TPB / HALVES;  // pixels per thread and half
constexpr int CHUNK = 8;                        // staging granule in pixels (16 B of depth)
#ifndef SLAMHIP_WIN_PITCH
#define SLAMHIP_WIN_PITCH 160
#endif
#ifndef SLAMHIP_WIN_ROWS
#define SLAMHIP_WIN_ROWS 24
#endif
#ifndef SLAMHIP_FUSED_WAVES
#define SLAMHIP_FUSED_WAVES 5   // minimum waves per SIMD the register allocation must allow
#endif
constexpr int WIN_PITCH = SLAMHIP_WIN_PITCH;    // LDS window row pitch in cells (multiple of CHUNK)
constexpr int WIN_ROWS = SLAMHIP_WIN_ROWS;
static_assert(STEPS * HALVES == 8 && TILE_W == 128 && TILE_H == 16, "pixel walk below: 2 x 64 columns, STEPS/2 rows per wave and half");

// workspace record of one (image, neighbour) pair: 64 bytes
struct PairAcc {
  double sums[SLAMHIP_NSUMS];
  unsigned int counts[SLAMHIP_NCOUNTS];
  unsigned int ticket;
  unsigned int pad[3];
};
static_assert(sizeof(PairAcc) == 64, "PairAcc must stay 64 bytes");

// DPP row/wave reductions: the result is valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// `old` (what masked-out rows receive) is the identity of the operation, which
// lets the compiler fold each step into one DPP-modified ALU instruction.
__device__ __forceinline__ float wave_sum63(float v) {
  v += dpp_f<0xB1, 0xF>(0.0f, v);         // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xF>(0.0f, v);         // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xF>(0.0f, v);        // row_half_mirror
  v += dpp_f<0x140, 0xF>(0.0f, v);        // row_mirror
  v += dpp_f<0x142, 0xA>(0.0f, v);        // row_bcast15 into rows 1,3
  v += dpp_f<0x143, 0xC>(0.0f, v);        // row_bcast31 into rows 2,3
  return v;
}
__device__ __forceinline__ int wave_min63(int v) {
  v = min(v, dpp_i<0xB1, 0xF>(INT_MAX, v));
  v = min(v, dpp_i<0x4E, 0xF>(INT_MAX, v));
  v = min(v, dpp_i<0x141, 0xF>(INT_MAX, v));
  v = min(v, dpp_i<0x140, 0xF>(INT_MAX, v));
  v = min(v, dpp_i<0x142, 0xA>(INT_MAX, v));
  v = min(v, dpp_i<0x143, 0xC>(INT_MAX, v));
  return v;
}
__device__ __forceinline__ int wave_max63(int v) {
  v = max(v, dpp_i<0xB1, 0xF>(INT_MIN, v));
  v = max(v, dpp_i<0x4E, 0xF>(INT_MIN, v));
  v = max(v, dpp_i<0x141, 0xF>(INT_MIN, v));
  v = max(v, dpp_i<0x140, 0xF>(INT_MIN, v));
  v = max(v, dpp_i<0x142, 0xA>(INT_MIN, v));
  v = max(v, dpp_i<0x143, 0xC>(INT_MIN, v));
  return v;
}

// Exact float64 division with the divisor-only work hoisted.  hipcc's IEEE a/b is
//   s = div_scale(b); n = div_scale(a); r = rcp(s); two Newton steps on r;
//   q = n*r; e = fma(-s, q, n); q = div_fmas(e, r, q); div_fixup(q, b, a)
// and both div_scale and div_fmas/div_fixup are identities when a and b are
// normal with exponents in [-300, 300] (no operand or quotient near the
// denormal or overflow range).  refined_rcp() + div_by() are the remaining
// instructions, so inside that band the quotient is bit-identical to a/b; the
// callers check the band and use plain a/b outside it.
__device__ __forceinline__ double refined_rcp(double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  e = fma(-b, r, 1.0);
  r = fma(r, e, r);
  return r;
}
__device__ __forceinline__ double div_by(double a, double b, double r) {
  const double q = a * r;
  const double e = fma(-b, q, a);
  return fma(e, r, q);
}
__device__ __forceinline__ bool in_band(double v) {  // 2^-300 <= |v| < 2^301
  const unsigned int hi = (unsigned int)__double2hiint(v) & 0x7fffffffu;
  return (hi - (723u << 20)) < (601u << 20);
}

// LDS cell of one neighbour-frame pixel (8 bytes):
//   x = half(1024 + R) | half(1024 + G) << 16,  y = half(1024 + B) | raw_depth << 16
// binary16 has ulp 1 on [1024, 2048), so half(1024 + n) is the bit pattern
// 0x6400 | n for a byte n: building a cell is byte shuffling, no conversion.
// Colour differences (exact packed-half subtractions) are free of the bias and
// each interpolation step is one mixed-precision fma; the bias rides along in
// the interpolated value and cancels against the equally biased anchor colour.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
constexpr unsigned int HALF_BIAS2 = 0x64006400u;   // two halves of 1024.0

__device__ __forceinline__ half2_t as_half2(unsigned int v) { return __builtin_bit_cast(half2_t, v); }
__device__ __forceinline__ uint2 make_cell(unsigned int r, unsigned int g, unsigned int b, unsigned int depth) {
  uint2 c;
  c.x = HALF_BIAS2 | r | (g << 16);
  c.y = 0x6400u | b | (depth << 16);
  return c;
}
__device__ __forceinline__ unsigned int pk_min_u16(unsigned int a, unsigned int b) {
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ unsigned int pk_max_u16(unsigned int a, unsigned int b) {
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
// packed (y << 16 | x) min / max over the first eight lanes, valid in lanes 0..7
__device__ __forceinline__ unsigned int oct_pk_min(unsigned int v) {
  v = pk_min_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(-1, (int)v));
  return pk_min_u16(v, (unsigned int)dpp_i<0x141, 0xF>(-1, (int)v));
}
__device__ __forceinline__ unsigned int oct_pk_max(unsigned int v) {
  v = pk_max_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(0, (int)v));
  return pk_max_u16(v, (unsigned int)dpp_i<0x141, 0xF>(0, (int)v));
}
// packed (y << 16 | x) min / max over the wave, valid in lane 63
__device__ __forceinline__ unsigned int wave_pk_min63(unsigned int v) {
  v = pk_min_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x141, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x140, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x142, 0xA>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x143, 0xC>(-1, (int)v));
  return v;
}
__device__ __forceinline__ unsigned int wave_pk_max63(unsigned int v) {
  v = pk_max_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x141, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x140, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x142, 0xA>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x143, 0xC>(0, (int)v));
  return v;
}

typedef float float2_t __attribute__((ext_vector_type(2)));

struct Acc {   // per-thread partial sums of one neighbour frame; the two lanes of each
  float2_t ap, sp, ag, sg;  // pair belong to even / odd steps and are added at the end
};

struct Cells {  // the four taps of one pixel
  uint2 a, b, c, e;
};

// Horizontal interpolation of one pixel: top / bottom rows of R, G, B and depth.
// Colour differences are exact in binary16; each lerp is one mixed-precision fma.
__device__ __forceinline__ void lerp_rows(const Cells& t, float fu, float top[4], float bot[4]) {
  const half2_t aRG = as_half2(t.a.x), cRG = as_half2(t.c.x);
  const half2_t dT = as_half2(t.b.x) - aRG, dB = as_half2(t.e.x) - cRG;
  const _Float16 aB = as_half2(t.a.y).x, cB = as_half2(t.c.y).x;
  const _Float16 dTb = as_half2(t.b.y).x - aB, dBb = as_half2(t.e.y).x - cB;
  top[0] = __builtin_fmaf(fu, (float)dT.x, (float)aRG.x);
  top[1] = __builtin_fmaf(fu, (float)dT.y, (float)aRG.y);
  top[2] = __builtin_fmaf(fu, (float)dTb, (float)aB);
  bot[0] = __builtin_fmaf(fu, (float)dB.x, (float)cRG.x);
  bot[1] = __builtin_fmaf(fu, (float)dB.y, (float)cRG.y);
  bot[2] = __builtin_fmaf(fu, (float)dBb, (float)cB);
  // depth, top and bottom row at once (packed float32: both lanes of an instruction are used)
  const float2_t d_ac = {(float)(t.a.y >> 16), (float)(t.c.y >> 16)}, d_be = {(float)(t.b.y >> 16), (float)(t.e.y >> 16)};
  const float2_t d_tb = __builtin_elementwise_fma(float2_t{fu, fu}, d_be - d_ac, d_ac);
  top[3] = d_tb.x;
  bot[3] = d_tb.y;
}

__device__ __forceinline__ bool taps_nonzero(const Cells& t) {  // all four raw depths != 0
  return min(min(t.a.y, t.b.y), min(t.c.y, t.e.y)) >= 0x10000u;
}

// Two pixels at once: everything after the horizontal lerps is packed float32
// (v_pk_add / v_pk_mul / v_pk_fma work on both pixels in one instruction).
// ok0/ok1: warp_valid.  Non-valid pixels have fu = fv = 0 and taps at cell 0, so
// their values are finite and a 0/1 factor removes them from the colour sums.
// Interpolation order as in the oracle (top, bottom, then vertical) with fused
// multiply-adds; the deviation from the unfused float32 oracle is a few 1e-7.
template <bool UNIT>
__device__ __forceinline__ unsigned int blend_accumulate2(Acc& acc, const Cells& t0, const Cells& t1, float fu0,
                                                          float fu1, float fv0, float fv1, bool ok0, bool ok1,
                                                          float factor_f, const unsigned int rgb0[3],
                                                          const unsigned int rgb1[3], double d0, double d1) {
  float top0[4], bot0[4], top1[4], bot1[4];
  lerp_rows(t0, fu0, top0, bot0);
  lerp_rows(t1, fu1, top1, bot1);
  const float2_t fv = {fv0, fv1};
  const float2_t okf = {ok0 ? 1.0f : 0.0f, ok1 ? 1.0f : 0.0f};
  float2_t w[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const float2_t top = {top0[c], top1[c]}, bot = {bot0[c], bot1[c]};
    w[c] = __builtin_elementwise_fma(fv, bot - top, top);
  }
  const float2_t rr = w[0] - float2_t{(float)rgb0[0], (float)rgb1[0]};
  const float2_t rg = w[1] - float2_t{(float)rgb0[1], (float)rgb1[1]};
  const float2_t rb = w[2] - float2_t{(float)rgb0[2], (float)rgb1[2]};
  const float2_t ap = {(fabsf(rr.x) + fabsf(rg.x)) + fabsf(rb.x), (fabsf(rr.y) + fabsf(rg.y)) + fabsf(rb.y)};
  const float2_t sp = __builtin_elementwise_fma(rb, rb, __builtin_elementwise_fma(rg, rg, rr * rr));
  acc.ap = __builtin_elementwise_fma(okf, ap, acc.ap);
  acc.sp = __builtin_elementwise_fma(okf, sp, acc.sp);
  float2_t wd = w[3];
  if (!UNIT) wd = wd / float2_t{factor_f, factor_f};
  const bool geo0 = ok0 && taps_nonzero(t0), geo1 = ok1 && taps_nonzero(t1);
  // depth residual in float64: d may be NaN/inf for a pixel that is not geo
  const float g0 = geo0 ? (float)((double)wd.x - d0) : 0.0f;
  const float g1 = geo1 ? (float)((double)wd.y - d1) : 0.0f;
  const float2_t g = {g0, g1};
  acc.ag += float2_t{fabsf(g0), fabsf(g1)};
  acc.sg = __builtin_elementwise_fma(g, g, acc.sg);
  return (geo0 ? 1u : 0u) | (geo1 ? 2u : 0u);
}

// Single-pixel form of blend_accumulate2 for the rare tiles that gather straight
// from global memory; kept small so that it does not set the kernel's register count.
// a_rg / a_b: the anchor's colour as biased binary16 patterns, half2(1024 + R, 1024 + G) and
// half(1024 + B) in the low 16 bits.  m_ok: lane mask of warp_valid; returns the lane mask of
// pixels that also have four non-zero depth taps.
template <bool UNIT>
__device__ __forceinline__ unsigned long long blend_accumulate1(Acc& acc, const Cells& t, float fu, float fv,
                                                                unsigned long long m_ok, float factor_f,
                                                                unsigned int a_rg, unsigned int a_b, double d) {
  float top[4], bot[4], w[4];
  lerp_rows(t, fu, top, bot);
  {   // vertical step: red and green in one packed instruction
    const float2_t t01 = {top[0], top[1]}, b01 = {bot[0], bot[1]};
    const float2_t w01 = __builtin_elementwise_fma(float2_t{fv, fv}, b01 - t01, t01);
    w[0] = w01.x;
    w[1] = w01.y;
    w[2] = __builtin_fmaf(fv, bot[2] - top[2], top[2]);
    w[3] = __builtin_fmaf(fv, bot[3] - top[3], top[3]);
  }
  // anchor - warped (the sign is irrelevant to |r| and r^2): one mixed-precision fma per channel,
  // the anchor (same bias of 1024 as the cells) read as binary16 in place
  float rr, rg, rb;
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[0,0,1]" : "=v"(rr) : "v"(w[0]), "v"(a_rg));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(rg) : "v"(w[1]), "v"(a_rg));
  asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[0,0,1]" : "=v"(rb) : "v"(w[2]), "v"(a_b));
  const float okf = __builtin_amdgcn_inverse_ballot_w64(m_ok) ? 1.0f : 0.0f;
  const float2_t as = {(fabsf(rr) + fabsf(rg)) + fabsf(rb), __builtin_fmaf(rb, rb, __builtin_fmaf(rg, rg, rr * rr))};
  const float2_t acc_as = __builtin_elementwise_fma(float2_t{okf, okf}, as, float2_t{acc.ap.x, acc.sp.x});
  acc.ap.x = acc_as.x;
  acc.sp.x = acc_as.y;
  const float wd = UNIT ? w[3] : w[3] / factor_f;
  const unsigned long long m_geo = m_ok & __builtin_amdgcn_ballot_w64(taps_nonzero(t));
  const float g = __builtin_amdgcn_inverse_ballot_w64(m_geo) ? (float)((double)wd - d) : 0.0f;
  acc.ag.x += fabsf(g);
  acc.sg.x = __builtin_fmaf(g, g, acc.sg.x);
  return m_geo;
}

__device__ __forceinline__ uint2 global_cell(const uint16_t* __restrict__ d2, const uint8_t* __restrict__ c2, int W,
                                             int x, int y) {
  const int o = y * W + x;
  return make_cell(c2[3 * o], c2[3 * o + 1], c2[3 * o + 2], d2[o]);
}

// Projection of one anchor pixel in float32, for window prediction only.
__device__ __forceinline__ void project_f32(const Cam& c, const Pose& p, float u0, float v0, float z, float& u,
                                            float& v, float& z2) {
  const float x = (u0 - (float)c.cx) * z * __builtin_amdgcn_rcpf((float)c.fx);
  const float y = (v0 - (float)c.cy) * z * __builtin_amdgcn_rcpf((float)c.fy);
  const float x2 = x * (float)p.r00 + y * (float)p.r10 + z * (float)p.r20 + (float)p.t0;
  const float y2 = x * (float)p.r01 + y * (float)p.r11 + z * (float)p.r21 + (float)p.t1;
  z2 = x * (float)p.r02 + y * (float)p.r12 + z * (float)p.r22 + (float)p.t2;
  const float rz = __builtin_amdgcn_rcpf(z2);
  u = x2 * (float)c.fx * rz + (float)c.cx;
  v = y2 * (float)c.fy * rz + (float)c.cy;
}

template <bool FAST, bool UNIT>
__global__ void __launch_bounds__(TPB, SLAMHIP_FUSED_WAVES)
pair_loss_fused_kernel(const uint16_t* __restrict__ depth1, const uint8_t* __restrict__ rgb1,
                       const uint16_t* __restrict__ depth2, const uint8_t* __restrict__ rgb2,
                       const double* __restrict__ rel, Cam cam, int B, int k, int H, int W,
                       int tiles_x, int tiles_y, int stage_vec, PairAcc* __restrict__ ws,
                       double* __restrict__ sums, unsigned int* __restrict__ counts) {
  __shared__ uint2 win[WIN_PITCH * WIN_ROWS];
  __shared__ unsigned int s_rng[4][2];
  __shared__ float s_sum[4][4];
  __shared__ unsigned int s_cnt[4][4];

  // blocks that share (blockIdx % 8) run on one XCD and share its L2: give each
  // such group whole images so that a tile's halo rows/columns hit in L2.
  const int tiles = tiles_x * tiles_y;
  const int group = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int b = (slot / tiles) * 8 + group;
  if (b >= B) return;
  const int tile = slot % tiles;
  const int tx = tile % tiles_x, ty = tile / tiles_x;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // known to be wave-uniform
  const long long HW = (long long)H * W;
  // step i of a thread = row (i>>1) of its wave's four rows, column (i&1)*64 + lane
  const int row0 = ty * TILE_H + wave * (STEPS / 2);
  const int colA = tx * TILE_W + lane;
  // wave-uniform plane bases + 32-bit per-lane offsets (clamped into the image)
  const uint16_t* d1b = depth1 + (long long)b * HW;
  const uint8_t* c1p = rgb1 + 3 * (long long)b * HW;
  // a step outside the image has no bit in the masks below; clamping the offset keeps its loads in bounds
  const int aoff0 = row0 * W + colA, alast = H * W - 1;
  auto anchor_offset = [&](int i) { return (unsigned int)min(aoff0 + (i >> 1) * W + (i & 1) * 64, alast); };

  const double dW = (double)W, dH = (double)H, dW1 = (double)(W - 1), dH1 = (double)(H - 1);
  const float factor_f = (float)cam.factor;
  // divisor-only parts of the three constant divisions (UNIT: factor == 1, z is the raw depth)
  const double r_fx = FAST ? refined_rcp(cam.fx) : 0.0;
  const double r_fy = FAST ? refined_rcp(cam.fy) : 0.0;
  const double r_fac = (FAST && !UNIT) ? refined_rcp(cam.factor) : 0.0;
  const double ucx0 = (double)colA - cam.cx, ucx1 = (double)(colA + 64) - cam.cx;
  // Lane masks of the step lying inside the image for the two columns, and a bit per row (wave-
  // uniform); FAST: and of "u - cx != 0" / "v - cy != 0", the per-column and per-row parts of valid0
  const unsigned long long col_nz0 = __builtin_amdgcn_ballot_w64((!FAST || ucx0 != 0.0) && colA < W);
  const unsigned long long col_nz1 = __builtin_amdgcn_ballot_w64((!FAST || ucx1 != 0.0) && colA + 64 < W);
  unsigned int rows_nz = 0u;
#pragma unroll
  for (int r = 0; r < STEPS / 2; ++r)
    rows_nz |= ((!FAST || __builtin_amdgcn_ballot_w64((double)(row0 + r) - cam.cy != 0.0) != 0ull) && row0 + r < H) ? (1u << r) : 0u;

  // range of non-zero anchor depth in the tile: found on first need (see below)
  unsigned int zmn = 0xffffu, zmx = 0u;
  bool have_range = false;

  for (int j = 0; j < k; ++j) {
    const long long bj = (long long)b * k + j;
    const Pose p = load_pose(rel + 16 * bj);
    const uint16_t* d2 = depth2 + bj * HW;
    const uint8_t* c2 = rgb2 + 3 * bj * HW;

    // ---- predicted tap window: the tile's corners at the nearest and farthest depth ----
    // For a fixed depth the tile maps to a convex quadrilateral, and a pixel moves
    // along a segment as its depth varies, so the eight projected points bound
    // every tap.  The window is only a cache: a tap outside it is gathered from
    // global memory, so the prediction (float32, with a margin) affects speed only.
    // When the translation moves a point at 256 raw depth units by less than half a
    // pixel, the window does not depend on depth and the tile's depth range is
    // not needed (a nearer pixel may miss the window and take the slow path).
    const float shift_px = fmaxf((float)cam.fx, (float)cam.fy) * factor_f * (1.0f / 256.0f) *
                           (fabsf((float)p.t0) + fabsf((float)p.t1) + fabsf((float)p.t2));
    const bool depth_free = shift_px < 0.5f;   // NaN: false
    if (!depth_free && !have_range) {
      // rolled on purpose: unrolled, its address arithmetic is hoisted out of the neighbour loop
      // and holds registers through the whole kernel for a branch that is rarely taken
#pragma unroll 1
      for (int i = 0; i < STEPS; ++i) {
        const bool inside = (row0 + (i >> 1) < H) && (colA + (i & 1) * 64 < W);
        const unsigned int dr = inside ? (unsigned int)d1b[anchor_offset(i)] : 0u;
        zmn = min(zmn, dr ? dr : 0xffffu);
        zmx = max(zmx, dr);
      }
      const unsigned int pk = wave_pk_min63(zmn | 0xffff0000u), px = wave_pk_max63(zmx);
      if (lane == 63) { s_rng[wave][0] = pk; s_rng[wave][1] = px; }
      __syncthreads();
      zmn = min(min(s_rng[0][0], s_rng[1][0]), min(s_rng[2][0], s_rng[3][0])) & 0xffffu;
      zmx = max(max(s_rng[0][1], s_rng[1][1]), max(s_rng[2][1], s_rng[3][1])) & 0xffffu;
      have_range = true;
    }
    const unsigned int z_near = depth_free ? 256u : zmn, z_far = depth_free ? 65535u : zmx;
    const bool any_depth = depth_free || zmx != 0u;
    int wx0 = 0, wy0 = 0, ncols = 0, nrows = 0;
    if (any_depth) {
      const int cx_ = (lane & 1) ? min(tx * TILE_W + TILE_W - 1, W - 1) : tx * TILE_W;
      const int cy_ = (lane & 2) ? min(ty * TILE_H + TILE_H - 1, H - 1) : ty * TILE_H;
      const float zc = (float)((lane & 4) ? z_far : z_near) * __builtin_amdgcn_rcpf(factor_f);
      float pu, pv, pz;
      project_f32(cam, p, (float)cx_, (float)cy_, zc, pu, pv, pz);
      const bool okp = (lane < 8) && (pz > 0.0f) && (fabsf(pu) < 1e6f) && (fabsf(pv) < 1e6f);   // NaN fails
      const bool bad = __builtin_amdgcn_ballot_w64((lane < 8) && !okp) != 0ull;
      const int lo_u = okp ? max((int)floorf(pu) - 2, 0) : 0x7fff, lo_v = okp ? max((int)floorf(pv) - 2, 0) : 0x7fff;
      const int hi_u = okp ? min((int)floorf(pu) + 3, W - 1) : 0, hi_v = okp ? min((int)floorf(pv) + 3, H - 1) : 0;
      const unsigned int pmin = oct_pk_min(((unsigned int)min(lo_v, 0x7fff) << 16) | (unsigned int)min(lo_u, 0xffff));
      const unsigned int pmax = oct_pk_max(((unsigned int)max(hi_v, 0) << 16) | (unsigned int)max(hi_u, 0));
      const unsigned int bmin = (unsigned int)__builtin_amdgcn_readfirstlane((int)pmin);
      const unsigned int bmax = (unsigned int)__builtin_amdgcn_readfirstlane((int)pmax);
      const int bx0 = (int)(bmin & 0xffffu), by0 = (int)(bmin >> 16), bx1 = (int)(bmax & 0xffffu), by1 = (int)(bmax >> 16);
      if (!bad && bx1 >= bx0 && by1 >= by0 && stage_vec) {
        wx0 = bx0 & ~(CHUNK - 1);
        wy0 = by0;
        ncols = min(bx1 + 1 - wx0, WIN_PITCH);   // a larger box is cut: the rest goes the slow way
        nrows = min(by1 + 1 - wy0, WIN_ROWS);
        if (ncols < 2 || nrows < 2) ncols = nrows = 0;   // a tap needs two columns and two rows
      }
    }
    // Pixels without taps in the window (not warp_valid, or missed) still read the
    // window's first 2 x 2 cells with zero weights: keep those cells finite.
    if (nrows == 0 && tid < 4) win[(tid >> 1) * WIN_PITCH + (tid & 1)] = make_uint2(0u, 0u);

    // ---- stage the window: 8-byte cells {half R | half G, half B | raw depth} ----
    const int chunks_x = (ncols + CHUNK - 1) / CHUNK;
    const int nchunks = chunks_x * nrows;
#ifdef SLAMHIP_ABL_NO_STAGE
    if (false)
#endif
    // chunk c -> (row, chunk in row) by a 16-bit reciprocal: exact for c < 2^16 / chunks_x
    const unsigned int inv_cx = chunks_x ? (65535u / (unsigned int)chunks_x + 1u) : 0u;
#pragma unroll 1
    for (int c = tid; c < nchunks; c += TPB) {
      const int cr = (int)(((unsigned int)c * inv_cx) >> 16), cc = c - cr * chunks_x;
      uint2* dst = win + cr * WIN_PITCH + cc * CHUNK;
      const int o = (wy0 + cr) * W + wx0 + cc * CHUNK;   // inside the image, 8-pixel aligned
      const uint4 dv = *reinterpret_cast<const uint4*>(d2 + o);
      const uint2* cp = reinterpret_cast<const uint2*>(c2 + 3 * o);
      const uint2 q0 = cp[0], q1 = cp[1], q2 = cp[2];
      const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
      const uint32_t cw[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, 0u};
#pragma unroll
      for (int i = 0; i < CHUNK; ++i) {
        // bytes 3i, 3i+1, 3i+2 of the 24-byte colour group; v_perm_b32 picks bytes
        // from {hi dword (4..7), lo dword (0..3)}, selector 0x0c gives a zero byte
        const int br = 3 * i, bg = 3 * i + 1, bb = 3 * i + 2;
        const int k = br >> 2;   // R and G lie in dwords k, k+1
        const unsigned int sel_rg = 0x0c000c00u | (unsigned int)(br - 4 * k) | ((unsigned int)(bg - 4 * k) << 16);
        const unsigned int sel_bd = 0x00000c00u | (unsigned int)(bb & 3) | ((4u + 2u * (i & 1)) << 16) | ((5u + 2u * (i & 1)) << 24);
        uint2 cell;
        cell.x = __builtin_amdgcn_perm(cw[k + 1], cw[k], sel_rg) | HALF_BIAS2;
        cell.y = __builtin_amdgcn_perm(dw[i >> 1], cw[bb >> 2], sel_bd) | 0x6400u;
        dst[i] = cell;
      }
    }
    __syncthreads();

    // ---- per pixel: project (float64), gather the four taps, blend, accumulate ----
    Acc acc = {{0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}, {0.f, 0.f}};
    unsigned int n_valid0 = 0, n_inb = 0, n_warp = 0, n_geo = 0;
    // a tap (du, dv) is in the window when du < lim_u and dv < lim_v (unsigned compares)
    const unsigned int lim_u = (unsigned int)max(chunks_x * CHUNK - 1, 0), lim_v = (unsigned int)max(nrows - 1, 0);
    // anchor depth and colour are (re)loaded one step ahead of their use; the
    // loop is deliberately not unrolled: one pixel's live set, small code
    unsigned int nxt[3];   // R | G << 8, B, raw depth
    {
      const unsigned int o = anchor_offset(0);
      const uint8_t* cp = c1p + 3u * o;
      nxt[0] = (unsigned int)cp[0] | ((unsigned int)cp[1] << 8); nxt[1] = cp[2]; nxt[2] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(d1b) + 2u * o);
    }
#ifndef SLAMHIP_ABL_NO_GATHER
#pragma unroll 1
    for (int i = 0; i < STEPS; ++i) {
      // bytes {R, 0x64, G, 0x64} = half2(1024 + R, 1024 + G); 0x6400 | B = half(1024 + B)
      const unsigned int a_rg = __builtin_amdgcn_perm(0x64u, nxt[0], 0x04010400u), a_b = 0x6400u | nxt[1];
      const unsigned int draw_i = nxt[2];   // a step outside the image has no bit in the valid0 masks
      {
        const unsigned int o = anchor_offset(min(i + 1, STEPS - 1));
        const uint8_t* cp = c1p + 3u * o;
        nxt[0] = (unsigned int)cp[0] | ((unsigned int)cp[1] << 8); nxt[1] = cp[2]; nxt[2] = *reinterpret_cast<const uint16_t*>(reinterpret_cast<const char*>(d1b) + 2u * o);
      }
      const double ucx = (i & 1) ? ucx1 : ucx0;
      const double vcy = (double)(row0 + (i >> 1)) - cam.cy;
      double x, y, z, x2, y2, z2, u, v;
      // Predicates live as 64-bit lane masks (one compare each, combined on the scalar unit)
      unsigned long long m_valid0;
      if (FAST) {
        // test_import.py:153-155 with the constant divisors' reciprocals hoisted
        z = UNIT ? (double)draw_i : div_by((double)draw_i, cam.factor, r_fac);
        x = div_by(ucx * z, cam.fx, r_fx);
        y = div_by(vcy * z, cam.fy, r_fy);
        // no product above can underflow (the host checked the constants), so
        // "x, y, z all non-zero" is "raw depth, u-cx and v-cy all non-zero"
        m_valid0 = __builtin_amdgcn_ballot_w64(draw_i != 0u) & ((i & 1) ? col_nz1 : col_nz0);
      } else {
        z = UNIT ? (double)draw_i : (double)draw_i / cam.factor;
        x = (ucx * z) / cam.fx;
        y = (vcy * z) / cam.fy;
        m_valid0 = __builtin_amdgcn_ballot_w64(x != 0.0) & __builtin_amdgcn_ballot_w64(y != 0.0) &
                   __builtin_amdgcn_ballot_w64(z != 0.0) & ((i & 1) ? col_nz1 : col_nz0);
      }
      if (!((rows_nz >> (i >> 1)) & 1u)) m_valid0 = 0ull;
      transform_px(p, x, y, z, x2, y2, z2);
      const double nu = x2 * cam.fx, nv = y2 * cam.fy;
      if (FAST) {
        // Only the divisor's range matters: with z2 normal and mid-range, a zero,
        // tiny, huge or non-finite numerator gives a quotient that is either exact
        // or (tiny) absorbed by "+ cx" / (huge, NaN) outside every window either
        // way, because the host admitted only |cx|, |cy| in [2^-32, 2^32].
        const unsigned long long m_band = __builtin_amdgcn_ballot_w64(in_band(z2));
        const double r_z = refined_rcp(z2);
        u = div_by(nu, z2, r_z) + cam.cx;
        v = div_by(nv, z2, r_z) + cam.cy;
        const unsigned long long m_plain = m_valid0 & ~m_band;
        if (m_plain != 0ull) {   // zero / huge / tiny / non-finite divisor: plain IEEE division
          if (__builtin_amdgcn_inverse_ballot_w64(m_plain)) {
            u = nu / z2 + cam.cx;
            v = nv / z2 + cam.cy;
          }
        }
      } else {
        u = nu / z2 + cam.cx;
        v = nv / z2 + cam.cy;
      }
      // test_import.py:174 (inclusive on both ends, NaN fails) and the tap window one pixel inside it
      const unsigned long long m_ge0 = m_valid0 & __builtin_amdgcn_ballot_w64(0.0 <= u) & __builtin_amdgcn_ballot_w64(0.0 <= v);
      const unsigned long long m_inb = m_ge0 & __builtin_amdgcn_ballot_w64(u <= dW) & __builtin_amdgcn_ballot_w64(v <= dH);
      const unsigned long long m_wv = m_ge0 & __builtin_amdgcn_ballot_w64(u <= dW1) & __builtin_amdgcn_ballot_w64(v <= dH1) &
                                      __builtin_amdgcn_ballot_w64(z2 > 0.0);
      n_valid0 += (unsigned int)__popcll(m_valid0);
      n_inb += (unsigned int)__popcll(m_inb);
      n_warp += (unsigned int)__popcll(m_wv);
      const bool wv = __builtin_amdgcn_inverse_ballot_w64(m_wv);
      // u, v >= 0 when warp_valid: truncation is floor and fract() is u - floor(u);
      // (the conversion instruction saturates and maps NaN to 0, so it runs on every lane;
      // only lanes of m_wv, where 0 <= u <= W - 1 and 0 <= v <= H - 1, use the index);
      // the weights of pixels that do not warp are zeroed (their u, v may be NaN/inf)
      int iu, iv;
      asm("v_cvt_i32_f64 %0, %1" : "=v"(iu) : "v"(u));
      asm("v_cvt_i32_f64 %0, %1" : "=v"(iv) : "v"(v));
      const float fu = wv ? (float)__builtin_amdgcn_fract(u) : 0.0f;
      const float fv = wv ? (float)__builtin_amdgcn_fract(v) : 0.0f;
      const int du = (int)((unsigned int)iu - (unsigned int)wx0), dv_ = (int)((unsigned int)iv - (unsigned int)wy0);   // may wrap in lanes outside m_wv
      const unsigned long long m_hit = m_wv & __builtin_amdgcn_ballot_w64((unsigned int)du < lim_u) &
                                       __builtin_amdgcn_ballot_w64((unsigned int)dv_ < lim_v);
      const bool hit = __builtin_amdgcn_inverse_ballot_w64(m_hit);
      const uint2* cell = win + (hit ? (unsigned int)dv_ * WIN_PITCH + (unsigned int)du : 0u);
      Cells t;
      t.a = cell[0]; t.b = cell[1]; t.c = cell[WIN_PITCH]; t.e = cell[WIN_PITCH + 1];
      const unsigned long long m_miss = m_wv & ~m_hit;
      if (m_miss != 0ull) {   // rare: taps outside the staged window
        if (__builtin_amdgcn_inverse_ballot_w64(m_miss)) {
          const int x1 = min(iu + 1, W - 1), y1 = min(iv + 1, H - 1);
          t.a = global_cell(d2, c2, W, iu, iv); t.b = global_cell(d2, c2, W, x1, iv);
          t.c = global_cell(d2, c2, W, iu, y1); t.e = global_cell(d2, c2, W, x1, y1);
        }
      }
      n_geo += (unsigned int)__popcll(blend_accumulate1<UNIT>(acc, t, fu, fv, m_wv, factor_f, a_rg, a_b, z2));
    }
#endif

    // ---- reduce: DPP within the wave, LDS across the four waves, one atomic each ----
    // (64 lanes adding to one LDS word serialise in the LDS pipe and stall the
    // wave far longer than the 6 DPP steps cost: measured 1.4x on the kernel)
    const float t0 = wave_sum63(acc.ap.x + acc.ap.y), t1 = wave_sum63(acc.sp.x + acc.sp.y);
    const float t2 = wave_sum63(acc.ag.x + acc.ag.y), t3 = wave_sum63(acc.sg.x + acc.sg.y);
    if (lane == 63) {
      s_sum[wave][0] = t0; s_sum[wave][1] = t1; s_sum[wave][2] = t2; s_sum[wave][3] = t3;
      s_cnt[wave][0] = n_valid0; s_cnt[wave][1] = n_inb; s_cnt[wave][2] = n_warp; s_cnt[wave][3] = n_geo;
    }
    __syncthreads();   // also: every gather is done before the next neighbour restages the window
    if (wave == 0) {
      PairAcc* pa = ws + bj;
      if (lane < 4) {
        const double t = ((double)s_sum[0][lane] + (double)s_sum[1][lane]) + ((double)s_sum[2][lane] + (double)s_sum[3][lane]);
        atomicAdd(&pa->sums[lane], t);
      } else if (lane < 8) {
        const int q = lane - 4;
        atomicAdd(&pa->counts[q], (s_cnt[0][q] + s_cnt[1][q]) + (s_cnt[2][q] + s_cnt[3][q]));
      }
#ifdef SLAMHIP_ABL_NO_TICKET
      continue;
#endif
      // every add of this workgroup has been performed before its ticket is drawn
      __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
      unsigned int t = 0;
      if (lane == 0) t = atomicAdd(&pa->ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t == (unsigned int)tiles - 1u) {
        // last workgroup of this pair: totals out, accumulators back to zero
        if (lane < 4) sums[bj * SLAMHIP_NSUMS + lane] = atomicExch(&pa->sums[lane], 0.0);
        else if (lane < 8) counts[bj * SLAMHIP_NCOUNTS + (lane - 4)] = atomicExch(&pa->counts[lane - 4], 0u);
        else if (lane == 8) atomicExch(&pa->ticket, 0u);
      }
    }
    if (j + 1 < k) __syncthreads();   // s_sum / s_cnt are read before the next neighbour rewrites them
  }
}

static bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

}  // namespace slamhip

using namespace slamhip;

extern "C" {

int slamhip_relative_pose_device(const float* gt7_a, const float* gt7_b, int n, double* rel4x4,
                                 void* stream) {
  SLAMHIP_REQUIRE(gt7_a && gt7_b && rel4x4 && n >= 0, SLAMHIP_E_ARG, "relative_pose_device: null pointer or n<0");
  if (n == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(relative_pose_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream, gt7_a,
                     gt7_b, n, rel4x4);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_backproject(const uint16_t* depth, int B, int H, int W, const double K[4], double factor,
                        void* xyz, int xyz_f64, uint8_t* valid0, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, SLAMHIP_E_ARG, "backproject: bad shape B=%d H=%d W=%d", B, H, W);
  const long long total = (long long)B * H * W;
  if (total == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(depth && K, SLAMHIP_E_ARG, "backproject: null depth or K");
  SLAMHIP_REQUIRE((total + 255) / 256 <= 0x7fffffffLL, SLAMHIP_E_SIZE, "backproject: too many pixels");
  const Cam cam{K[0], K[1], K[2], K[3], factor};
  const dim3 grid((unsigned)((total + 255) / 256)), block(256);
  if (xyz_f64)
    hipLaunchKernelGGL(backproject_kernel<double>, grid, block, 0, (hipStream_t)stream, depth, cam, H, W,
                       total, (double*)xyz, valid0);
  else
    hipLaunchKernelGGL(backproject_kernel<float>, grid, block, 0, (hipStream_t)stream, depth, cam, H, W,
                       total, (float*)xyz, valid0);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_reproject(const void* xyz, int xyz_f64, const uint8_t* valid0, const double* rel4x4,
                      const double K[4], double dim_u, double dim_v, int B, int H, int W, void* xyz2,
                      void* uvd, int out_f64, int32_t* iu_iv, uint8_t* inb, uint32_t* counts,
                      void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && H > 0 && W > 0, SLAMHIP_E_ARG, "reproject: bad shape B=%d H=%d W=%d", B, H, W);
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(xyz && K, SLAMHIP_E_ARG, "reproject: null xyz or K");
  hipStream_t s = (hipStream_t)stream;
  if (counts && B > 0) SLAMHIP_HIP(hipMemsetAsync(counts, 0, sizeof(uint32_t) * 2 * (size_t)B, s));
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE((long long)H * W <= 0x7fffff00LL, SLAMHIP_E_SIZE, "reproject: image too large");
  const int HW = H * W;
  const int bpi = (HW + 255) / 256;
  SLAMHIP_REQUIRE((long long)bpi * B <= 0x7fffffffLL, SLAMHIP_E_SIZE, "reproject: too many pixels");
  const Cam cam{K[0], K[1], K[2], K[3], 1.0};
  const dim3 grid((unsigned)((long long)bpi * B)), block(256);
#define SLAMHIP_LAUNCH_REPROJECT(IN, OUT)                                                               \
  hipLaunchKernelGGL((reproject_kernel<IN, OUT>), grid, block, 0, s, (const IN*)xyz, valid0, rel4x4, cam, \
                     dim_u, dim_v, HW, bpi, (OUT*)xyz2, (OUT*)uvd, iu_iv, inb, counts)
  if (xyz_f64 && out_f64) SLAMHIP_LAUNCH_REPROJECT(double, double);
  else if (xyz_f64) SLAMHIP_LAUNCH_REPROJECT(double, float);
  else if (out_f64) SLAMHIP_LAUNCH_REPROJECT(float, double);
  else SLAMHIP_LAUNCH_REPROJECT(float, float);
#undef SLAMHIP_LAUNCH_REPROJECT
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_overlap_count(const double* uvd, long long n, double dim_u, double dim_v, uint8_t* mask,
                          uint32_t* count, void* stream) {
  SLAMHIP_REQUIRE(uvd && count && n >= 0, SLAMHIP_E_ARG, "overlap_count: null pointer or n<0");
  hipStream_t s = (hipStream_t)stream;
  SLAMHIP_HIP(hipMemsetAsync(count, 0, sizeof(uint32_t), s));
  if (n == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE((n + 255) / 256 <= 0x7fffffffLL, SLAMHIP_E_SIZE, "overlap_count: too many points");
  hipLaunchKernelGGL(overlap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, uvd, n, dim_u,
                     dim_v, mask, count);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

size_t slamhip_pair_loss_workspace_bytes(int B, int k) {
  if (B <= 0 || k <= 0) return 0;
  return sizeof(PairAcc) * (size_t)B * (size_t)k;
}

// Argument checks and launch shared by the two fused entry points; `workspace` holds at least
// B * k zero-filled PairAcc records.
static int launch_pair_loss(const char* who, const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                            const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor, int B,
                            int k, int H, int W, double* sums, uint32_t* counts, void* workspace, hipStream_t s) {
  SLAMHIP_REQUIRE(depth1 && rgb1 && depth2 && rgb2 && rel4x4 && K && sums && counts && workspace, SLAMHIP_E_ARG,
                  "%s: null pointer", who);
  SLAMHIP_REQUIRE(H <= 32767 && W <= 65535 && (long long)H * W <= 0x20000000LL, SLAMHIP_E_SIZE,
                  "%s: image larger than 65535 x 32767 or 2^29 pixels", who);
  SLAMHIP_REQUIRE(aligned_to(workspace, 8) && aligned_to(sums, 8) && aligned_to(counts, 4), SLAMHIP_E_ARG,
                  "%s: misaligned workspace or output", who);
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  const long long groups = ((long long)B + 7) / 8;
  const long long nblocks = groups * 8 * tiles_x * tiles_y;
  SLAMHIP_REQUIRE(nblocks <= 0x7fffffffLL, SLAMHIP_E_SIZE, "%s: grid too large", who);
  const Cam cam{K[0], K[1], K[2], K[3], factor};
  // the window staging uses 16-byte / 8-byte vector loads: needs W % 8 == 0 and aligned planes
  const int stage_vec = (W % CHUNK == 0) && aligned_to(depth2, 16) && aligned_to(rgb2, 8);
  // hoisted-reciprocal divisions are exact while nothing comes near the denormal
  // or overflow range: constants in [2^-64, 2^64], principal point in [2^-32, 2^32]
  auto mid = [](double v) { const double a = fabs(v); return a >= 0x1p-64 && a <= 0x1p64; };
  auto pp = [](double v) { const double a = fabs(v); return a >= 0x1p-32 && a <= 0x1p32; };
  const bool fast = mid(K[0]) && mid(K[1]) && mid(factor) && pp(K[2]) && pp(K[3]);
  PairAcc* ws = (PairAcc*)workspace;
  const bool unit = factor == 1.0;
#define SLAMHIP_LAUNCH_FUSED(F, U)                                                                         \
  hipLaunchKernelGGL((pair_loss_fused_kernel<F, U>), dim3((unsigned)nblocks), dim3(TPB), 0, s, depth1, rgb1, \
                     depth2, rgb2, rel4x4, cam, B, k, H, W, tiles_x, tiles_y, stage_vec, ws, sums, counts)
  if (fast && unit) SLAMHIP_LAUNCH_FUSED(true, true);
  else if (fast) SLAMHIP_LAUNCH_FUSED(true, false);
  else if (unit) SLAMHIP_LAUNCH_FUSED(false, true);
  else SLAMHIP_LAUNCH_FUSED(false, false);
#undef SLAMHIP_LAUNCH_FUSED
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_pair_loss_fused(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                            const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor,
                            int B, int k, int H, int W, double* sums, uint32_t* counts, void* workspace,
                            size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && k > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "pair_loss_fused: bad shape B=%d k=%d H=%d W=%d", B, k, H, W);
  if (B == 0) return SLAMHIP_OK;  // an empty batch has no buffers to check
  SLAMHIP_REQUIRE(workspace != nullptr, SLAMHIP_E_ARG, "pair_loss_fused: null pointer");
  SLAMHIP_REQUIRE(workspace_bytes >= slamhip_pair_loss_workspace_bytes(B, k), SLAMHIP_E_SIZE,
                  "pair_loss_fused: workspace of %zu bytes, need %zu", workspace_bytes,
                  slamhip_pair_loss_workspace_bytes(B, k));
  return launch_pair_loss("pair_loss_fused", depth1, rgb1, depth2, rgb2, rel4x4, K, factor, B, k, H, W, sums,
                          counts, workspace, (hipStream_t)stream);
}

int slamhip_pair_loss_reset_workspace(void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(workspace || workspace_bytes == 0, SLAMHIP_E_ARG, "pair_loss_reset_workspace: null workspace");
  if (workspace_bytes == 0) return SLAMHIP_OK;
  SLAMHIP_HIP(hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream));
  return SLAMHIP_OK;
}

// ---- the fused kernel for callers that cannot allocate ---------------------
// The accumulator records live in a file-scope device array instead of a caller's
// workspace.  Every GPU holds its own copy of the array at its own address: the
// address is looked up once per device (with that device current) and only that
// pointer, never the array's name, is used by host code afterwards.
static __device__ __align__(256) unsigned char scratch[SLAMHIP_STATIC_MAX_PAIRS * sizeof(PairAcc)];
constexpr int MAX_DEVICES = 64;
static std::atomic<void*> scratch_ptr[MAX_DEVICES];

int slamhip_pair_loss_fused_static(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                                   const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor,
                                   int B, int k, int H, int W, double* sums, uint32_t* counts, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && k > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "pair_loss_fused_static: bad shape B=%d k=%d H=%d W=%d", B, k, H, W);
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE((long long)B * k <= SLAMHIP_STATIC_MAX_PAIRS, SLAMHIP_E_SIZE,
                  "pair_loss_fused_static: %lld (anchor, neighbour) pairs, the built-in scratch holds %d "
                  "(use slamhip_pair_loss_fused with a workspace)", (long long)B * k, SLAMHIP_STATIC_MAX_PAIRS);
  int dev = -1;
  SLAMHIP_HIP(hipGetDevice(&dev));
  SLAMHIP_REQUIRE(dev >= 0 && dev < MAX_DEVICES, SLAMHIP_E_SIZE, "pair_loss_fused_static: device %d", dev);
  void* ptr = scratch_ptr[dev].load(std::memory_order_acquire);
  if (!ptr) {   // first call on this device (racing first calls look up the same address)
    SLAMHIP_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    SLAMHIP_REQUIRE(ptr != nullptr, SLAMHIP_E_HIP, "pair_loss_fused_static: no address for the scratch array");
    scratch_ptr[dev].store(ptr, std::memory_order_release);
  }
  hipStream_t s = (hipStream_t)stream;
  // the array keeps whatever the last call (possibly an interrupted one) left: clear the records of this call
  SLAMHIP_HIP(hipMemsetAsync(ptr, 0, sizeof(PairAcc) * (size_t)B * (size_t)k, s));
  return launch_pair_loss("pair_loss_fused_static", depth1, rgb1, depth2, rgb2, rel4x4, K, factor, B, k, H, W,
                          sums, counts, ptr, s);
}

int slamhip_warp_materialize(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                             const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor,
                             int B, int k, int H, int W, float* warped_rgb, float* warped_depth,
                             float* r_photo, float* r_geo, uint8_t* mask, int32_t* iu_iv, double* uvd,
                             void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && k > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "warp_materialize: bad shape B=%d k=%d H=%d W=%d", B, k, H, W);
  const long long total = (long long)B * k * H * W;
  if (total == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(depth1 && rgb1 && depth2 && rgb2 && rel4x4 && K, SLAMHIP_E_ARG,
                  "warp_materialize: null pointer");
  SLAMHIP_REQUIRE((total + 255) / 256 <= 0x7fffffffLL, SLAMHIP_E_SIZE, "warp_materialize: too many pixels");
  const Cam cam{K[0], K[1], K[2], K[3], factor};
  hipLaunchKernelGGL(warp_materialize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, depth1, rgb1, depth2, rgb2, rel4x4, cam, k, H, W, total, warped_rgb,
                     warped_depth, r_photo, r_geo, mask, iu_iv, uvd);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

}  // extern "C"
