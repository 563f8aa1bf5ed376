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

#include <climits>
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
  return (0.0 <= u) && (u <= du) && (0.0 <= v) && (v <= dv);
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
// One 256-thread workgroup owns a TILE_W x TILE_H block of anchor pixels; a
// thread owns PX consecutive pixels of one row (one 16-byte depth load, three
// 8-byte colour loads).  Per neighbour frame:
//   A  project the PX pixels (float64), keep (iu, iv, fu, fv, d, flags);
//   B  workgroup bounding box of the bilinear taps;
//   C  stage that window of the neighbour frame (colour + depth) into LDS as
//      8-byte cells {R|G<<8|B<<16, float(depth)} with 16-byte global loads;
//   D  gather the four taps of every pixel from LDS, blend, accumulate;
//   a window wider than WIN_PITCH or taller than WIN_ROWS (large parallax)
//   falls back to gathering straight from global memory for that tile.
// Sums are reduced by wave shuffles, then across the four waves through LDS,
// then one float64 / uint32 atomic per quantity per workgroup.
constexpr int PX = 8;
constexpr int TILE_W = 128;
constexpr int TILE_H = 16;
constexpr int TPB = (TILE_W / PX) * TILE_H;   // 256
constexpr int WIN_PITCH = 160;                // LDS window row pitch in cells (multiple of PX)
constexpr int WIN_ROWS = 30;
static_assert(TPB == 256, "tile must map to 256 threads");

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

template <bool VEC>
__global__ void __launch_bounds__(TPB)
pair_loss_fused_kernel(const uint16_t* __restrict__ depth1, const uint8_t* __restrict__ rgb1,
                       const uint16_t* __restrict__ depth2, const uint8_t* __restrict__ rgb2,
                       const double* __restrict__ rel, Cam cam, int B, int k, int H, int W,
                       int tiles_x, int tiles_y, double* __restrict__ sums,
                       unsigned int* __restrict__ counts) {
  __shared__ uint2 win[WIN_PITCH * WIN_ROWS];
  __shared__ int s_box[4][4];
  __shared__ float s_sum[4][4];
  __shared__ unsigned int s_cnt[4][4];

  // blocks that share (blockIdx % 8) run on one XCD and share its L2: give each
  // such group whole images so a tile's halo rows/columns hit in L2.
  const int tiles = tiles_x * tiles_y;
  const int group = blockIdx.x & 7;
  const int slot = blockIdx.x >> 3;
  const int b = (slot / tiles) * 8 + group;
  if (b >= B) return;
  const int tile = slot % tiles;
  const int tx = tile % tiles_x, ty = tile / tiles_x;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int row = ty * TILE_H + tid / (TILE_W / PX);
  const int col0 = tx * TILE_W + (tid % (TILE_W / PX)) * PX;
  const long long HW = (long long)H * W;
  const bool row_ok = row < H;

  // ---- anchor pixels: raw depth and colour of PX pixels ----
  uint32_t draw[PX];
  uint32_t crgb[PX];  // R | G<<8 | B<<16
#pragma unroll
  for (int i = 0; i < PX; ++i) { draw[i] = 0; crgb[i] = 0; }
  if (row_ok) {
    const long long o = (long long)b * HW + (long long)row * W + col0;
    if (VEC && col0 + PX <= W) {
      const uint4 dv = *reinterpret_cast<const uint4*>(depth1 + o);
      const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
#pragma unroll
      for (int i = 0; i < PX; ++i) draw[i] = (dw[i >> 1] >> ((i & 1) * 16)) & 0xffffu;
      const uint2* cp = reinterpret_cast<const uint2*>(rgb1 + 3 * o);
      const uint2 c0 = cp[0], c1 = cp[1], c2 = cp[2];
      const uint32_t cw[6] = {c0.x, c0.y, c1.x, c1.y, c2.x, c2.y};
#pragma unroll
      for (int i = 0; i < PX; ++i) {
        const int bo = 3 * i;  // byte offset of pixel i in the 24-byte group
        const uint64_t two = (uint64_t)cw[bo >> 2] | ((uint64_t)cw[min((bo >> 2) + 1, 5)] << 32);
        crgb[i] = (uint32_t)(two >> ((bo & 3) * 8)) & 0xffffffu;
      }
    } else {
#pragma unroll
      for (int i = 0; i < PX; ++i)
        if (col0 + i < W) {
          draw[i] = depth1[o + i];
          const uint8_t* c = rgb1 + 3 * (o + i);
          crgb[i] = (uint32_t)c[0] | ((uint32_t)c[1] << 8) | ((uint32_t)c[2] << 16);
        }
    }
  }

  const double dW = (double)W, dH = (double)H, dW1 = (double)(W - 1), dH1 = (double)(H - 1);
  const float factor_f = (float)cam.factor;

  for (int j = 0; j < k; ++j) {
    const long long bj = (long long)b * k + j;
    const Pose p = load_pose(rel + 16 * bj);
    const uint16_t* d2 = depth2 + bj * HW;
    const uint8_t* c2 = rgb2 + 3 * bj * HW;

    // ---- A: project ----
    int iu[PX], iv[PX];
    float fu[PX], fv[PX];
    double dd[PX];
    unsigned int flags = 0;  // bit i: warp_valid of pixel i
    unsigned int n_valid0 = 0, n_inb = 0, n_warp = 0, n_geo = 0;
    int bx0 = INT_MAX, bx1 = INT_MIN, by0 = INT_MAX, by1 = INT_MIN;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      double x, y, z, x2, y2, z2, u, v;
      backproject_px(cam, col0 + i, row, draw[i], x, y, z);
      transform_px(p, x, y, z, x2, y2, z2);
      project_px(cam, x2, y2, z2, u, v);
      const bool live = row_ok && (col0 + i < W);
      const bool valid0 = live && (x != 0.0 && y != 0.0 && z != 0.0);
      const bool inb = valid0 && in_window(u, v, dW, dH);
      const bool wv = valid0 && in_window(u, v, dW1, dH1) && (z2 > 0.0);
      n_valid0 += valid0; n_inb += inb; n_warp += wv;
      const double fl_u = floor(u), fl_v = floor(v);
      iu[i] = wv ? (int)fl_u : 0;
      iv[i] = wv ? (int)fl_v : 0;
      fu[i] = (float)(u - fl_u);
      fv[i] = (float)(v - fl_v);
      dd[i] = z2;
      if (wv) {
        flags |= 1u << i;
        bx0 = min(bx0, iu[i]); bx1 = max(bx1, iu[i]);
        by0 = min(by0, iv[i]); by1 = max(by1, iv[i]);
      }
    }

    // ---- B: workgroup bounding box of the taps ----
    bx0 = wave_min(bx0); by0 = wave_min(by0); bx1 = wave_max(bx1); by1 = wave_max(by1);
    if (lane == 0) { s_box[wave][0] = bx0; s_box[wave][1] = by0; s_box[wave][2] = bx1; s_box[wave][3] = by1; }
    __syncthreads();
    bx0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0]));
    by0 = min(min(s_box[0][1], s_box[1][1]), min(s_box[2][1], s_box[3][1]));
    bx1 = max(max(s_box[0][2], s_box[1][2]), max(s_box[2][2], s_box[3][2]));
    by1 = max(max(s_box[0][3], s_box[1][3]), max(s_box[2][3], s_box[3][3]));
    const bool any = bx1 >= bx0;                 // some pixel of the tile warps
    const int wx0 = any ? (bx0 & ~(PX - 1)) : 0; // window origin, aligned to the staging chunk
    const int wy0 = any ? by0 : 0;
    const int ncols = any ? (bx1 + 2 - wx0) : 0; // taps reach iu+1 / iv+1
    const int nrows = any ? (by1 + 2 - wy0) : 0;
    const bool use_lds = VEC && any && ncols <= WIN_PITCH && nrows <= WIN_ROWS;

    // ---- C: stage the window ----
    if (use_lds) {
      const int chunks_x = (ncols + PX - 1) / PX;
      const int nchunks = chunks_x * nrows;
      for (int c = tid; c < nchunks; c += TPB) {
        const int cr = c / chunks_x, cc = c - cr * chunks_x;
        const int sy = min(wy0 + cr, H - 1);        // clamp-to-edge rows
        const int sx = wx0 + cc * PX;
        uint2* dst = win + cr * WIN_PITCH + cc * PX;
        if (sx + PX <= W) {
          const long long o = (long long)sy * W + sx;
          const uint4 dv = *reinterpret_cast<const uint4*>(d2 + o);
          const uint2* cp = reinterpret_cast<const uint2*>(c2 + 3 * o);
          const uint2 q0 = cp[0], q1 = cp[1], q2 = cp[2];
          const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
          const uint32_t cw[6] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y};
#pragma unroll
          for (int i = 0; i < PX; ++i) {
            const int bo = 3 * i;
            const uint64_t two = (uint64_t)cw[bo >> 2] | ((uint64_t)cw[min((bo >> 2) + 1, 5)] << 32);
            uint2 cell;
            cell.x = (uint32_t)(two >> ((bo & 3) * 8)) & 0xffffffu;
            cell.y = __float_as_uint((float)((dw[i >> 1] >> ((i & 1) * 16)) & 0xffffu));
            dst[i] = cell;
          }
        } else {
          // chunk past the right edge: replicate the last column (clamp-to-edge)
          const long long o = (long long)sy * W + (W - 1);
          uint2 cell;
          cell.x = (uint32_t)c2[3 * o] | ((uint32_t)c2[3 * o + 1] << 8) | ((uint32_t)c2[3 * o + 2] << 16);
          cell.y = __float_as_uint((float)d2[o]);
#pragma unroll
          for (int i = 0; i < PX; ++i) dst[i] = cell;
        }
      }
    }
    __syncthreads();

    // ---- D: gather, blend, accumulate ----
    float s_ap = 0.f, s_sp = 0.f, s_ag = 0.f, s_sg = 0.f;
#pragma unroll
    for (int i = 0; i < PX; ++i) {
      if (flags & (1u << i)) {
        Tap a, bb, c, e;
        if (use_lds) {
          const uint2* cell = win + (iv[i] - wy0) * WIN_PITCH + (iu[i] - wx0);
          a = unpack_cell(cell[0]);
          bb = unpack_cell(cell[1]);
          c = unpack_cell(cell[WIN_PITCH]);
          e = unpack_cell(cell[WIN_PITCH + 1]);
        } else {
          const int x1 = min(iu[i] + 1, W - 1), y1 = min(iv[i] + 1, H - 1);
          a = global_tap(d2, c2, W, iu[i], iv[i]);
          bb = global_tap(d2, c2, W, x1, iv[i]);
          c = global_tap(d2, c2, W, iu[i], y1);
          e = global_tap(d2, c2, W, x1, y1);
        }
        const WarpOut o = blend(a, bb, c, e, fu[i], fv[i], factor_f);
        const float r0 = o.wr - (float)(crgb[i] & 0xffu);
        const float r1 = o.wg - (float)((crgb[i] >> 8) & 0xffu);
        const float r2 = o.wb - (float)((crgb[i] >> 16) & 0xffu);
        s_ap += (fabsf(r0) + fabsf(r1)) + fabsf(r2);
        s_sp += (r0 * r0 + r1 * r1) + r2 * r2;
        if (o.geo) {
          const float rg = (float)((double)o.wd - dd[i]);
          s_ag += fabsf(rg);
          s_sg += rg * rg;
          n_geo += 1;
        }
      }
    }

    // ---- reduce: wave shuffles, then the four waves, then one atomic each ----
    s_ap = wave_sum(s_ap); s_sp = wave_sum(s_sp); s_ag = wave_sum(s_ag); s_sg = wave_sum(s_sg);
    const unsigned int c0 = (unsigned int)wave_sum((float)n_valid0), c1 = (unsigned int)wave_sum((float)n_inb);
    const unsigned int c2n = (unsigned int)wave_sum((float)n_warp), c3 = (unsigned int)wave_sum((float)n_geo);
    if (lane == 0) {
      s_sum[wave][0] = s_ap; s_sum[wave][1] = s_sp; s_sum[wave][2] = s_ag; s_sum[wave][3] = s_sg;
      s_cnt[wave][0] = c0; s_cnt[wave][1] = c1; s_cnt[wave][2] = c2n; s_cnt[wave][3] = c3;
    }
    __syncthreads();
    if (tid < 4) {
      const double t = ((double)s_sum[0][tid] + (double)s_sum[1][tid]) + ((double)s_sum[2][tid] + (double)s_sum[3][tid]);
      atomicAdd(&sums[bj * SLAMHIP_NSUMS + tid], t);
    } else if (tid < 8) {
      const int q = tid - 4;
      atomicAdd(&counts[bj * SLAMHIP_NCOUNTS + q], (s_cnt[0][q] + s_cnt[1][q]) + (s_cnt[2][q] + s_cnt[3][q]));
    }
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

int slamhip_pair_loss_fused(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                            const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor,
                            int B, int k, int H, int W, double* sums, uint32_t* counts, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && k > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "pair_loss_fused: bad shape B=%d k=%d H=%d W=%d", B, k, H, W);
  if (B == 0) return SLAMHIP_OK;  // an empty batch has no buffers to check
  SLAMHIP_REQUIRE(depth1 && rgb1 && depth2 && rgb2 && rel4x4 && K && sums && counts, SLAMHIP_E_ARG,
                  "pair_loss_fused: null pointer");
  hipStream_t s = (hipStream_t)stream;
  SLAMHIP_HIP(hipMemsetAsync(sums, 0, sizeof(double) * SLAMHIP_NSUMS * (size_t)B * k, s));
  SLAMHIP_HIP(hipMemsetAsync(counts, 0, sizeof(uint32_t) * SLAMHIP_NCOUNTS * (size_t)B * k, s));
  const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
  const long long groups = ((long long)B + 7) / 8;
  const long long nblocks = groups * 8 * tiles_x * tiles_y;
  SLAMHIP_REQUIRE(nblocks <= 0x7fffffffLL, SLAMHIP_E_SIZE, "pair_loss_fused: grid too large");
  const Cam cam{K[0], K[1], K[2], K[3], factor};
  // the 16-byte / 8-byte vector loads need W % 8 == 0 and 16-byte aligned planes
  const bool vec = (W % PX == 0) && aligned_to(depth1, 16) && aligned_to(depth2, 16) &&
                   aligned_to(rgb1, 8) && aligned_to(rgb2, 8);
  if (vec)
    hipLaunchKernelGGL(pair_loss_fused_kernel<true>, dim3((unsigned)nblocks), dim3(TPB), 0, s, depth1, rgb1,
                       depth2, rgb2, rel4x4, cam, B, k, H, W, tiles_x, tiles_y, sums, counts);
  else
    hipLaunchKernelGGL(pair_loss_fused_kernel<false>, dim3((unsigned)nblocks), dim3(TPB), 0, s, depth1, rgb1,
                       depth2, rgb2, rel4x4, cam, B, k, H, W, tiles_x, tiles_y, sums, counts);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
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
