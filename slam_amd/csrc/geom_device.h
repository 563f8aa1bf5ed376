// Device-side pieces shared by the per-pixel geometry kernels (geometry.hip: the
// materialising kernels; pair_loss.hip: the fused frame-pair kernel): camera and
// pose records and the reference's per-pixel arithmetic in its own operation order.
#pragma once
#include <hip/hip_runtime.h>

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

// Counters that kernels add into are cleared by this kernel on the stream, not by hipMemsetAsync: a sequence of
// calls recorded in a HIP graph then holds kernel nodes only.
static __global__ void __launch_bounds__(256) clear_u32_kernel(unsigned int* __restrict__ p, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}
static inline void clear_u32_async(unsigned int* p, long long n, hipStream_t s) {
  hipLaunchKernelGGL(clear_u32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, p, n);
}

// floor to int32, saturating, NaN -> INT32_MIN
__device__ __forceinline__ int32_t pixel_index(double c) {
  if (c != c) return INT_MIN;
  const double f = floor(c);
  if (f <= -2147483648.0) return INT_MIN;
  if (f >= 2147483647.0) return INT_MAX;
  return (int32_t)f;
}

}  // namespace slamhip
