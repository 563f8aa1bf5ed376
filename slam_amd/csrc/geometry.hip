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
// This file holds the materialising kernels (per-pixel outputs: the facade's results and the
// independent cross-check of the fused kernel in pair_loss.hip).  They are HBM-bound streaming
// kernels with a float64 VALU load; nothing here is GEMM-shaped, so no MFMA.
#include "geom_device.h"

namespace slamhip {

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

constexpr int REPROJECT_PX = 8;   // pixels a thread of reproject_kernel walks
template <typename InT, typename OutT>
__global__ void __launch_bounds__(256)
reproject_kernel(const InT* __restrict__ xyz, const uint8_t* __restrict__ valid0,
                 const double* __restrict__ rel, Cam cam, double dim_u, double dim_v, int HW,
                 int blocks_per_image, OutT* __restrict__ xyz2, OutT* __restrict__ uvd,
                 int32_t* __restrict__ iu_iv, uint8_t* __restrict__ inb,
                 unsigned int* __restrict__ counts) {
  // a workgroup never straddles two images, so the pose is wave-uniform; it walks REPROJECT_PX pixels per thread
  // and adds its counts with one atomic pair (counters that every wave of an image adds to by itself serialise:
  // 3.4 ms for 64 frames of 640 x 480 with counts alone)
  __shared__ unsigned int s_n[4][2];
  const int b = blockIdx.x / blocks_per_image;
  const int px0 = (blockIdx.x - b * blocks_per_image) * (256 * REPROJECT_PX) + threadIdx.x;
  unsigned int n0 = 0u, n1 = 0u;   // wave-uniform
#pragma unroll 1
  for (int it = 0; it < REPROJECT_PX; ++it) {
    const int px = px0 + 256 * it;
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
    n0 += (unsigned int)__popcll(__ballot(v0));
    n1 += (unsigned int)__popcll(__ballot(ib));
  }
  if (counts) {
    if ((threadIdx.x & 63) == 0) { s_n[threadIdx.x >> 6][0] = n0; s_n[threadIdx.x >> 6][1] = n1; }
    __syncthreads();
    if (threadIdx.x < 2) {
      const unsigned int c = (s_n[0][threadIdx.x] + s_n[1][threadIdx.x]) + (s_n[2][threadIdx.x] + s_n[3][threadIdx.x]);
      if (c) atomicAdd(&counts[2 * b + threadIdx.x], c);
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
  if (counts && B > 0) clear_u32_async(counts, 2LL * B, s);
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE((long long)H * W <= 0x7fffff00LL, SLAMHIP_E_SIZE, "reproject: image too large");
  const int HW = H * W;
  const int bpi = (HW + 256 * REPROJECT_PX - 1) / (256 * REPROJECT_PX);
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
  clear_u32_async(count, 1, s);
  if (n == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE((n + 255) / 256 <= 0x7fffffffLL, SLAMHIP_E_SIZE, "overlap_count: too many points");
  hipLaunchKernelGGL(overlap_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, uvd, n, dim_u,
                     dim_v, mask, count);
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
