// Input preparation of the RGB-D trainers on the GPU (slam/network/model_input.py:427-450
// and the per-batch mean subtraction of :360-362, :517-522): per frame
//   depth -> 255 * depth / max(depth)   (integer division, as the Python-2 reference)
//   concat with RGB to 4 channels, bilinear resize to Ho x Wo, float32
// then the per-batch channel mean is subtracted.  The resize restates
// skimage.transform.resize(order=1) as interpolation at half-pixel centres with edge
// clamping (the library is not available to pin it; the host implementation in
// slam_amd/network/model_input.py uses the same definition).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"

namespace slamhip {

typedef __bf16 bf16_t;

// the workspace (sums, per-frame maxima) is cleared by a kernel, not by hipMemsetAsync: calls recorded in a HIP
// graph then hold kernel nodes only
__global__ void __launch_bounds__(256) clear_words_kernel(unsigned int* __restrict__ p, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] = 0u;
}

typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int pk_max_u16(unsigned int a, unsigned int b) {
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(u16x2_t, a), __builtin_bit_cast(u16x2_t, b)));
}

// max of each frame's raw depth (uint32 per frame, zeroed by the caller): 16-byte loads over the part of the
// frame that starts at a 16-byte boundary (packed 16-bit maxima), single samples before and after it; one
// atomic per workgroup
__global__ void __launch_bounds__(256)
depth_max_kernel(const uint16_t* __restrict__ depth, unsigned int* __restrict__ dmax, long long hw) {
  __shared__ unsigned int part[4];
  const int b = blockIdx.y;
  const uint16_t* d = depth + (long long)b * hw;
  const long long head = min(hw, (long long)(((16u - (unsigned int)(reinterpret_cast<uintptr_t>(d) & 15u)) & 15u) >> 1));
  const long long n16 = (hw - head) >> 3;
  const uint4* q = reinterpret_cast<const uint4*>(d + head);
  unsigned int m2 = 0u;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (long long)gridDim.x * blockDim.x) {
    const uint4 v = q[i];
    m2 = pk_max_u16(pk_max_u16(m2, v.x), pk_max_u16(v.y, pk_max_u16(v.z, v.w)));
  }
  unsigned int m = max(m2 & 0xffffu, m2 >> 16);
  if (blockIdx.x == 0) {
    for (long long i = threadIdx.x; i < head; i += blockDim.x) m = max(m, (unsigned int)d[i]);
    for (long long i = head + 8 * n16 + threadIdx.x; i < hw; i += blockDim.x) m = max(m, (unsigned int)d[i]);
  }
  for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned int)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(part[0], part[1]), max(part[2], part[3]));
    if (m) atomicMax(&dmax[b], m);
  }
}

// A workgroup covers 32 output columns x 32 output rows of one frame; a thread walks RS_ROWS consecutive rows
// of one column (its column's source coordinates are made once).  4 channels, float64 interpolation, float32
// result; per-channel sums of the batch for the mean (float64; one atomic per channel and workgroup, the
// workgroups spread over two sets of sums).  The depth taps' integer quotient 255 d / max is
// floor((255 d + 0.5) * (1 / max)) in float64: exact, because a quotient that is no integer lies at least
// 1 / max >= 2^-16 from the next one and the product is good to 2^-28.
constexpr int RS_ROWS = 4;
__global__ void __launch_bounds__(256)
resize_rgbd_kernel(const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth,
                   const unsigned int* __restrict__ dmax, float* __restrict__ out, double* __restrict__ sums, int B,
                   int Hs, int Ws, int Ho, int Wo) {
  __shared__ double part[4][4];
  const int b = blockIdx.z;
  const int xo = blockIdx.x * 32 + (threadIdx.x & 31);
  const int yo0 = (blockIdx.y * 8 + (threadIdx.x >> 5)) * RS_ROWS;
  double v[4] = {0.0, 0.0, 0.0, 0.0};
  if (xo < Wo) {
    double xs = ((double)xo + 0.5) * (double)Ws / (double)Wo - 0.5;
    xs = fmin(fmax(xs, 0.0), (double)(Ws - 1));
    const int x0 = (int)floor(xs);
    const int x1 = min(x0 + 1, Ws - 1);
    const double fx = xs - (double)x0;
    const unsigned int plane = (unsigned int)b * (unsigned int)(Hs * Ws);   // (B Hs Ws < 2^32: checked by the host)
    const double rmax = 1.0 / (double)max(dmax[b], 1u);
    auto scaled = [&](unsigned int raw) { return (double)(unsigned int)(((double)(255u * raw) + 0.5) * rmax); };
#pragma unroll
    for (int j = 0; j < RS_ROWS; ++j) {
      const int yo = yo0 + j;
      if (yo >= Ho) break;
      double ys = ((double)yo + 0.5) * (double)Hs / (double)Ho - 0.5;
      ys = fmin(fmax(ys, 0.0), (double)(Hs - 1));
      const int y0 = (int)floor(ys);
      const int y1 = min(y0 + 1, Hs - 1);
      const double fy = ys - (double)y0;
      const unsigned int r0 = plane + (unsigned int)(y0 * Ws), r1 = plane + (unsigned int)(y1 * Ws);
      const unsigned int o00 = r0 + x0, o01 = r0 + x1, o10 = r1 + x0, o11 = r1 + x1;
      const uint8_t *p00 = rgb + 3ull * o00, *p01 = rgb + 3ull * o01, *p10 = rgb + 3ull * o10, *p11 = rgb + 3ull * o11;
      float4 res;
      float* rp = &res.x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        double a, bq, cq, e;
        if (c < 3) {
          a = p00[c]; bq = p01[c]; cq = p10[c]; e = p11[c];
        } else {
          a = scaled(depth[o00]); bq = scaled(depth[o01]); cq = scaled(depth[o10]); e = scaled(depth[o11]);
        }
        const double top = a * (1.0 - fx) + bq * fx, bot = cq * (1.0 - fx) + e * fx;
        rp[c] = (float)(top * (1.0 - fy) + bot * fy);
        v[c] += (double)rp[c];     // the mean is taken over the float32 images
      }
      reinterpret_cast<float4*>(out)[((long long)b * Ho + yo) * Wo + xo] = res;
    }
  }
  // block sums -> one atomic per channel
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double s = v[c];
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][c] = s;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    const int set = (blockIdx.x + blockIdx.y + blockIdx.z) & 1;
    atomicAdd(&sums[4 * set + threadIdx.x], (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
  }
}

// out -= mean (float32 arithmetic on a float64 mean, like numpy's float32 - float64 -> float64 -> cast)
__global__ void __launch_bounds__(256)
subtract_mean_kernel(float* __restrict__ out, bf16_t* __restrict__ out_bf16, const double* __restrict__ sums,
                     double* __restrict__ mean_out, long long total4, double inv_count) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && mean_out) {
#pragma unroll
    for (int c = 0; c < 4; ++c) mean_out[c] = (sums[c] + sums[4 + c]) * inv_count;
  }
  if (i >= total4) return;
  float4 v = reinterpret_cast<float4*>(out)[i];
  v.x = (float)((double)v.x - (sums[0] + sums[4]) * inv_count);
  v.y = (float)((double)v.y - (sums[1] + sums[5]) * inv_count);
  v.z = (float)((double)v.z - (sums[2] + sums[6]) * inv_count);
  v.w = (float)((double)v.w - (sums[3] + sums[7]) * inv_count);
  reinterpret_cast<float4*>(out)[i] = v;
  if (out_bf16) {
    out_bf16[4 * i] = (bf16_t)v.x; out_bf16[4 * i + 1] = (bf16_t)v.y;
    out_bf16[4 * i + 2] = (bf16_t)v.z; out_bf16[4 * i + 3] = (bf16_t)v.w;
  }
}

// The in-graph variant (model_input.py:185-217): the depth PNG decoded to 8 bits (the
// upper byte of the 16-bit sample), both images resized with TensorFlow 0.x's bilinear
// rule (source coordinate = index * in / out, no half-pixel shift, float32 lerp), then
// depth * 255 / max(resized depth); no mean subtraction.  Pass 1 writes the resized
// frame and the per-frame maximum (non-negative floats order like their bit patterns).
__global__ void __launch_bounds__(256)
resize_rgbd_tf_kernel(const uint8_t* __restrict__ rgb, const uint16_t* __restrict__ depth, float* __restrict__ out,
                      unsigned int* __restrict__ dmax_bits, int B, int Hs, int Ws, int Ho, int Wo) {
  // a workgroup covers 32 output columns x 32 output rows of one frame, a thread RS_ROWS consecutive rows of one
  // column; the frame's maximum takes one atomic per workgroup
  __shared__ unsigned int part[4];
  const int b = blockIdx.z;
  const int xo = blockIdx.x * 32 + (threadIdx.x & 31);
  const int yo0 = (blockIdx.y * 8 + (threadIdx.x >> 5)) * RS_ROWS;
  unsigned int mbits = 0u;
  if (xo < Wo) {
    const float in_x = (float)xo * ((float)Ws / (float)Wo);
    const int x0 = (int)floorf(in_x);
    const int x1 = min((int)ceilf(in_x), Ws - 1);
    const float fx = in_x - (float)x0;
    const unsigned int plane = (unsigned int)b * (unsigned int)(Hs * Ws);   // (B Hs Ws < 2^32: checked by the host)
#pragma unroll
    for (int j = 0; j < RS_ROWS; ++j) {
      const int yo = yo0 + j;
      if (yo >= Ho) break;
      const float in_y = (float)yo * ((float)Hs / (float)Ho);
      const int y0 = (int)floorf(in_y);
      const int y1 = min((int)ceilf(in_y), Hs - 1);
      const float fy = in_y - (float)y0;
      const unsigned int r0 = plane + (unsigned int)(y0 * Ws), r1 = plane + (unsigned int)(y1 * Ws);
      const unsigned int o00 = r0 + x0, o01 = r0 + x1, o10 = r1 + x0, o11 = r1 + x1;
      const uint8_t *p00 = rgb + 3ull * o00, *p01 = rgb + 3ull * o01, *p10 = rgb + 3ull * o10, *p11 = rgb + 3ull * o11;
      float4 res;
      float* rp = &res.x;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        float a, bq, cq, e;
        if (c < 3) {
          a = p00[c]; bq = p01[c]; cq = p10[c]; e = p11[c];
        } else {
          a = (float)(depth[o00] >> 8); bq = (float)(depth[o01] >> 8);
          cq = (float)(depth[o10] >> 8); e = (float)(depth[o11] >> 8);
        }
        const float top = a + (bq - a) * fx, bot = cq + (e - cq) * fx;
        rp[c] = top + (bot - top) * fy;
      }
      reinterpret_cast<float4*>(out)[((long long)b * Ho + yo) * Wo + xo] = res;
      mbits = max(mbits, __float_as_uint(fmaxf(res.w, 0.0f)));   // (non-negative floats order like their bit patterns)
    }
  }
  for (int o = 32; o > 0; o >>= 1) mbits = max(mbits, (unsigned int)__shfl_xor((int)mbits, o));
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mbits;
  __syncthreads();
  if (threadIdx.x == 0) {
    mbits = max(max(part[0], part[1]), max(part[2], part[3]));
    if (mbits) atomicMax(&dmax_bits[b], mbits);
  }
}

// grid: (blocks over a frame's output pixels, frames)
__global__ void __launch_bounds__(256)
scale_depth_tf_kernel(float* __restrict__ out, const unsigned int* __restrict__ dmax_bits, int hw) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hw) return;
  const float mx = __uint_as_float(dmax_bits[blockIdx.y]);
  float* o = out + 4 * ((long long)blockIdx.y * hw + i) + 3;
  *o = *o * 255.0f / mx;     // an all-zero depth image gives NaN, as in the reference
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

size_t slamhip_prep_rgbd_workspace_bytes(int B) { return B > 0 ? 64 + sizeof(unsigned int) * (size_t)B : 0; }

int slamhip_prep_rgbd(const uint8_t* rgb, const uint16_t* depth, int B, int Hs, int Ws, int Ho, int Wo, float* out_f32,
                      void* out_bf16, double* mean4, void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, SLAMHIP_E_ARG, "prep_rgbd: bad shape");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(rgb && depth && out_f32 && workspace, SLAMHIP_E_ARG, "prep_rgbd: null pointer");
  SLAMHIP_REQUIRE(workspace_bytes >= slamhip_prep_rgbd_workspace_bytes(B), SLAMHIP_E_SIZE,
                  "prep_rgbd: workspace of %zu bytes, need %zu", workspace_bytes, slamhip_prep_rgbd_workspace_bytes(B));
  SLAMHIP_REQUIRE((long long)Hs * Ws <= 0x7fffffffLL && (long long)B * Ho * Wo <= 0x7fffffffLL &&
                      (long long)B * Hs * Ws <= 0xffffffffLL && B <= 65535,
                  SLAMHIP_E_SIZE, "prep_rgbd: too many pixels or frames (at most 2^32 source pixels and 65535 frames per call)");
  hipStream_t s = (hipStream_t)stream;
  double* sums = (double*)workspace;                              // two sets of 4 doubles (64 bytes)
  unsigned int* dmax = (unsigned int*)((char*)workspace + 64);    // B
  {
    const long long words = (long long)(slamhip_prep_rgbd_workspace_bytes(B) / 4);
    hipLaunchKernelGGL(clear_words_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (unsigned int*)workspace, words);
  }
  const long long hw = (long long)Hs * Ws;
  hipLaunchKernelGGL(depth_max_kernel, dim3((unsigned)((hw + 4095) / 4096 < 64 ? (hw + 4095) / 4096 : 64), B), dim3(256), 0,
                     s, depth, dmax, hw);
  const long long total = (long long)B * Ho * Wo;
  hipLaunchKernelGGL(resize_rgbd_kernel, dim3((unsigned)((Wo + 31) / 32), (unsigned)((Ho + 8 * RS_ROWS - 1) / (8 * RS_ROWS)), (unsigned)B),
                     dim3(256), 0, s, rgb, depth, dmax, out_f32, sums, B, Hs, Ws, Ho, Wo);
  hipLaunchKernelGGL(subtract_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, out_f32,
                     (bf16_t*)out_bf16, sums, mean4, total, 1.0 / (double)total);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_prep_rgbd_tf(const uint8_t* rgb, const uint16_t* depth, int B, int Hs, int Ws, int Ho, int Wo,
                         float* out_f32, void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && Ho > 0 && Wo > 0, SLAMHIP_E_ARG, "prep_rgbd_tf: bad shape");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(rgb && depth && out_f32 && workspace, SLAMHIP_E_ARG, "prep_rgbd_tf: null pointer");
  SLAMHIP_REQUIRE(workspace_bytes >= slamhip_prep_rgbd_workspace_bytes(B), SLAMHIP_E_SIZE,
                  "prep_rgbd_tf: workspace of %zu bytes, need %zu", workspace_bytes, slamhip_prep_rgbd_workspace_bytes(B));
  SLAMHIP_REQUIRE((long long)Hs * Ws <= 0x7fffffffLL && (long long)B * Ho * Wo <= 0x7fffffffLL &&
                      (long long)B * Hs * Ws <= 0xffffffffLL && B <= 65535,
                  SLAMHIP_E_SIZE, "prep_rgbd_tf: too many pixels or frames (at most 2^32 source pixels and 65535 frames per call)");
  hipStream_t s = (hipStream_t)stream;
  unsigned int* dmax = (unsigned int*)((char*)workspace + 64);
  {
    const long long words = (long long)(slamhip_prep_rgbd_workspace_bytes(B) / 4);
    hipLaunchKernelGGL(clear_words_kernel, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, s, (unsigned int*)workspace, words);
  }
  hipLaunchKernelGGL(resize_rgbd_tf_kernel, dim3((unsigned)((Wo + 31) / 32), (unsigned)((Ho + 8 * RS_ROWS - 1) / (8 * RS_ROWS)), (unsigned)B),
                     dim3(256), 0, s, rgb, depth, out_f32, dmax, B, Hs, Ws, Ho, Wo);
  hipLaunchKernelGGL(scale_depth_tf_kernel, dim3((unsigned)((Ho * Wo + 255) / 256), (unsigned)B), dim3(256), 0, s, out_f32, dmax,
                     Ho * Wo);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

}  // extern "C"
