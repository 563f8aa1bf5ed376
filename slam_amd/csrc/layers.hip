// Layer kernels of the GoogLeNet / PoseNet encoder (slam/network/google_net.py:14-215)
// around the matrix-core GEMMs of encoder.hip.  The layer semantics are those of the
// kaffe.tensorflow Network helpers the reference builds on (an un-vendored dependency:
// conv = tf.nn.conv2d SAME + bias + ReLU, max/avg pool SAME unless stated, LRN =
// tf.nn.local_response_normalization, concat on the channel axis, fc = x.W + b).
//
// A convolution of any window and stride runs as
//   im2col (bf16 [M][Kp], K = KH*KW*Cin in (ky, kx, ci) order, zero padded to Kp % 64 == 0)
//   -> slamhip_gemm_nt_bf16 -> bias/ReLU epilogue that writes into a channel slice of the
//   destination (inception branches write straight into their concat buffer),
// and backwards as slice epilogue (dz, dbias) -> slamhip_wgrad_bf16 on the same im2col
// matrix -> GEMM with the untransposed filter -> col2im gather.  Every input-gradient
// kernel here ADDS into its destination: a tensor that feeds several branches collects
// their gradients in a buffer the caller zeroes once per step.
//
// SAME padding follows TensorFlow: out = ceil(in / stride), total padding
// max((out - 1) * stride + k - in, 0), the smaller half in front.  Activations are NHWC
// bfloat16 with channel counts that are multiples of 8 (16-byte accesses).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.h"
#include "zero_fill.h"

namespace slamhip {

typedef __bf16 bf16_t;

__device__ __forceinline__ float bf16_bits_to_float(unsigned int b) { return __uint_as_float(b << 16); }

static inline unsigned grid_for(long long n, int threads = 256) { return (unsigned)((n + threads - 1) / threads); }

struct ConvGeom {
  int B, H, W, C;         // input
  int KH, KW, stride;
  int Ho, Wo, pad_t, pad_l;
  int K, Kp;              // KH*KW*C and its padded width
};

__global__ void __launch_bounds__(256)
im2col_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, ConvGeom g, long long chunks) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chunks) return;
  const int per_row = g.Kp >> 3;
  const long long m = idx / per_row;
  const int kk = (int)(idx - m * per_row) << 3;
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if (kk < g.K) {
    const int tap = kk / g.C, c = kk - tap * g.C;
    const int ky = tap / g.KW, kx = tap - ky * g.KW;
    const int ox = (int)(m % g.Wo);
    const int oy = (int)((m / g.Wo) % g.Ho);
    const int b = (int)(m / ((long long)g.Wo * g.Ho));
    const int iy = oy * g.stride + ky - g.pad_t, ix = ox * g.stride + kx - g.pad_l;
    if (iy >= 0 && iy < g.H && ix >= 0 && ix < g.W)
      v = *reinterpret_cast<const uint4*>(x + (((long long)b * g.H + iy) * g.W + ix) * g.C + c);
  }
  *reinterpret_cast<uint4*>(col + m * g.Kp + kk) = v;
}

// dx[b, iy, ix, c..c+8) += sum over the windows that cover (iy, ix) of dcol[m][tap * C + c..]
__global__ void __launch_bounds__(256)
col2im_add_kernel(const bf16_t* __restrict__ dcol, bf16_t* __restrict__ dx, ConvGeom g, long long chunks) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= chunks) return;
  const int per_px = g.C >> 3;
  const long long px = idx / per_px;
  const int c = (int)(idx - px * per_px) << 3;
  const int ix = (int)(px % g.W);
  const int iy = (int)((px / g.W) % g.H);
  const int b = (int)(px / ((long long)g.W * g.H));
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int ky = 0; ky < g.KH; ++ky) {
    const int ny = iy + g.pad_t - ky;
    if (ny < 0 || ny % g.stride) continue;
    const int oy = ny / g.stride;
    if (oy >= g.Ho) continue;
    for (int kx = 0; kx < g.KW; ++kx) {
      const int nx = ix + g.pad_l - kx;
      if (nx < 0 || nx % g.stride) continue;
      const int ox = nx / g.stride;
      if (ox >= g.Wo) continue;
      const long long m = ((long long)b * g.Ho + oy) * g.Wo + ox;
      const uint4 v = *reinterpret_cast<const uint4*>(dcol + m * g.Kp + (ky * g.KW + kx) * g.C + c);
      const unsigned int w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += bf16_bits_to_float(w[j] & 0xffffu);
        acc[2 * j + 1] += bf16_bits_to_float(w[j] >> 16);
      }
    }
  }
  bf16_t* o = dx + px * g.C + c;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16_t)((float)o[j] + acc[j]);
}

// y[m * ld + off + n] = act(z[m * Np + n] + bias[n]),  n < N
__global__ void __launch_bounds__(256)
bias_act_slice_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias, bf16_t* __restrict__ y,
                          float* __restrict__ y_f32, long long total, int N, int Np, int ld, int off, int relu) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const long long m = idx / N;
  const int n = (int)(idx - m * N);
  float v = z[m * Np + n] + bias[n];
  if (relu) v = fmaxf(v, 0.0f);
  if (y) y[m * ld + off + n] = (bf16_t)v;
  if (y_f32) y_f32[m * ld + off + n] = v;
}

// dz[m][n] = dy[m * ld + off + n] * (y > 0 or 1) for n < N, 0 for N <= n < Np; dbias[n] += column sums
__global__ void __launch_bounds__(256)
bias_act_slice_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ dy_f32, const bf16_t* __restrict__ y,
                          bf16_t* __restrict__ dz, float* __restrict__ dbias, long long M, int N, int Np, int ld, int off,
                          int relu, long long rows_per_block) {
  const int n = blockIdx.x * 64 + (threadIdx.x & 63);
  const int sub = threadIdx.x >> 6;                       // 4 row phases per workgroup
  const long long m0 = (long long)blockIdx.y * rows_per_block;
  const long long m1 = m0 + rows_per_block < M ? m0 + rows_per_block : M;
  if (n >= Np) return;
  float s = 0.0f;
  for (long long m = m0 + sub; m < m1; m += 4) {
    float v = 0.0f;
    if (n < N) {
      const long long o = m * ld + off + n;
      v = dy ? (float)dy[o] : dy_f32[o];
      if (relu && !((float)y[o] > 0.0f)) v = 0.0f;
    }
    const bf16_t r = (bf16_t)v;
    dz[m * Np + n] = r;
    s += (float)r;
  }
  if (n < N && s != 0.0f) atomicAdd(&dbias[n], s);
}

struct PoolGeom {
  int B, H, W, C;
  int K, stride;
  int Ho, Wo, pad_t, pad_l;
  int is_max;
};

// one thread per output element; average pooling divides by the number of window cells inside the image
__global__ void __launch_bounds__(256)
pool_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, PoolGeom g, long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % g.C);
  const long long m = idx / g.C;
  const int ox = (int)(m % g.Wo);
  const int oy = (int)((m / g.Wo) % g.Ho);
  const int b = (int)(m / ((long long)g.Wo * g.Ho));
  float best = -INFINITY, sum = 0.0f;
  int cnt = 0;
  for (int ky = 0; ky < g.K; ++ky) {
    const int iy = oy * g.stride + ky - g.pad_t;
    if (iy < 0 || iy >= g.H) continue;
    for (int kx = 0; kx < g.K; ++kx) {
      const int ix = ox * g.stride + kx - g.pad_l;
      if (ix < 0 || ix >= g.W) continue;
      const float v = (float)x[(((long long)b * g.H + iy) * g.W + ix) * g.C + c];
      best = fmaxf(best, v);
      sum += v;
      ++cnt;
    }
  }
  y[idx] = (bf16_t)(g.is_max ? best : sum / (float)cnt);
}

// one thread per input element: dx += sum over the windows that cover it.  Max pooling gives a
// window's gradient to its first maximum in (ky, kx) order.
__global__ void __launch_bounds__(256)
pool_bwd_add_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, PoolGeom g,
                    long long total) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % g.C);
  const long long px = idx / g.C;
  const int ix = (int)(px % g.W);
  const int iy = (int)((px / g.W) % g.H);
  const int b = (int)(px / ((long long)g.W * g.H));
  const float mine = g.is_max ? (float)x[idx] : 0.0f;
  float acc = 0.0f;
  for (int ky = 0; ky < g.K; ++ky) {
    const int ny = iy + g.pad_t - ky;
    if (ny < 0 || ny % g.stride) continue;
    const int oy = ny / g.stride;
    if (oy >= g.Ho) continue;
    for (int kx = 0; kx < g.K; ++kx) {
      const int nx = ix + g.pad_l - kx;
      if (nx < 0 || nx % g.stride) continue;
      const int ox = nx / g.stride;
      if (ox >= g.Wo) continue;
      const float gy = (float)dy[(((long long)b * g.Ho + oy) * g.Wo + ox) * g.C + c];
      // walk this window: am I its first maximum (max) / how many cells does it hold (average)
      bool first_max = true;
      int cnt = 0;
      for (int wy = 0; wy < g.K; ++wy) {
        const int jy = oy * g.stride + wy - g.pad_t;
        if (jy < 0 || jy >= g.H) continue;
        for (int wx = 0; wx < g.K; ++wx) {
          const int jx = ox * g.stride + wx - g.pad_l;
          if (jx < 0 || jx >= g.W) continue;
          ++cnt;
          if (g.is_max) {
            const float v = (float)x[(((long long)b * g.H + jy) * g.W + jx) * g.C + c];
            const bool before = wy < ky || (wy == ky && wx < kx);
            if (v > mine || (before && v == mine)) first_max = false;
          }
        }
      }
      acc += g.is_max ? (first_max ? gy : 0.0f) : gy / (float)cnt;
    }
  }
  if (acc != 0.0f) dx[idx] = (bf16_t)((float)dx[idx] + acc);
}

// y = x / (bias + alpha * sum_{|j - c| <= r} x_j^2)^beta over the channel axis
__global__ void __launch_bounds__(256)
lrn_fwd_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, long long total, int C, int radius, float bias,
               float alpha, float beta) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const bf16_t* row = x + (idx - c);
  float s = 0.0f;
  for (int j = max(c - radius, 0); j <= min(c + radius, C - 1); ++j) {
    const float v = (float)row[j];
    s = fmaf(v, v, s);
  }
  y[idx] = (bf16_t)((float)row[c] * powf(bias + alpha * s, -beta));
}

// dx_c += dy_c s_c^-beta - 2 alpha beta x_c sum_{|j - c| <= r} dy_j x_j s_j^(-beta-1)
__global__ void __launch_bounds__(256)
lrn_bwd_add_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, long long total,
                   int C, int radius, float bias, float alpha, float beta) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % C);
  const bf16_t* row = x + (idx - c);
  const bf16_t* grow = dy + (idx - c);
  float cross = 0.0f, own = 0.0f;
  for (int j = max(c - radius, 0); j <= min(c + radius, C - 1); ++j) {
    float s = 0.0f;
    for (int i = max(j - radius, 0); i <= min(j + radius, C - 1); ++i) {
      const float v = (float)row[i];
      s = fmaf(v, v, s);
    }
    const float base = bias + alpha * s;
    const float p = powf(base, -beta);
    if (j == c) own = (float)grow[c] * p;
    cross += (float)grow[j] * (float)row[j] * p / base;
  }
  dx[idx] = (bf16_t)((float)dx[idx] + own - 2.0f * alpha * beta * (float)row[c] * cross);
}

// float32 [P][C] -> bfloat16 [P][Cp], channels C..Cp-1 zero
__global__ void __launch_bounds__(256)
pad_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long long total, int C, int Cp) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int c = (int)(idx % Cp);
  const long long p = idx / Cp;
  dst[idx] = (bf16_t)(c < C ? src[p * C + c] : 0.0f);
}

static bool same_geometry(int in, int k, int stride, int same, int* out, int* pad_front) {
  if (in <= 0 || k <= 0 || stride <= 0) return false;
  if (same) {
    *out = (in + stride - 1) / stride;
    const int total = (*out - 1) * stride + k - in;
    *pad_front = total > 0 ? total / 2 : 0;
  } else {
    if (in < k) return false;
    *out = (in - k) / stride + 1;
    *pad_front = 0;
  }
  return true;
}

static bool conv_geometry(int B, int H, int W, int C, int KH, int KW, int stride, ConvGeom* g) {
  g->B = B; g->H = H; g->W = W; g->C = C; g->KH = KH; g->KW = KW; g->stride = stride;
  if (!same_geometry(H, KH, stride, 1, &g->Ho, &g->pad_t) || !same_geometry(W, KW, stride, 1, &g->Wo, &g->pad_l))
    return false;
  g->K = KH * KW * C;
  g->Kp = (g->K + 63) / 64 * 64;
  return true;
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

int slamhip_conv_out_shape(int H, int W, int KH, int KW, int stride, int same, int* Ho, int* Wo) {
  int pt, pl;
  SLAMHIP_REQUIRE(Ho && Wo && same_geometry(H, KH, stride, same, Ho, &pt) && same_geometry(W, KW, stride, same, Wo, &pl),
                  SLAMHIP_E_ARG, "conv_out_shape: bad argument");
  return SLAMHIP_OK;
}

int slamhip_im2col_bf16(const void* x, void* col, int B, int H, int W, int C, int KH, int KW, int stride, void* stream) {
  ConvGeom g;
  SLAMHIP_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && conv_geometry(B, H, W, C, KH, KW, stride, &g), SLAMHIP_E_ARG,
                  "im2col: bad shape (channels must be a multiple of 8)");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(x && col, SLAMHIP_E_ARG, "im2col: null pointer");
  const long long M = (long long)B * g.Ho * g.Wo;
  SLAMHIP_REQUIRE(M <= 0x7fffffffLL, SLAMHIP_E_SIZE, "im2col: too many output pixels");
  const long long chunks = M * (g.Kp / 8);
  hipLaunchKernelGGL(im2col_kernel, dim3(grid_for(chunks)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)col, g, chunks);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_col2im_add_bf16(const void* dcol, void* dx, int B, int H, int W, int C, int KH, int KW, int stride,
                            void* stream) {
  ConvGeom g;
  SLAMHIP_REQUIRE(B >= 0 && C > 0 && C % 8 == 0 && conv_geometry(B, H, W, C, KH, KW, stride, &g), SLAMHIP_E_ARG,
                  "col2im_add: bad shape (channels must be a multiple of 8)");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(dcol && dx, SLAMHIP_E_ARG, "col2im_add: null pointer");
  const long long chunks = (long long)B * H * W * (C / 8);
  hipLaunchKernelGGL(col2im_add_kernel, dim3(grid_for(chunks)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)dcol, (bf16_t*)dx, g, chunks);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bias_act_slice_fwd(const float* z, const float* bias, void* y_bf16, float* y_f32, long long M, int N, int Np,
                               int ld, int off, int relu, void* stream) {
  SLAMHIP_REQUIRE(M >= 0 && N > 0 && Np >= N && ld >= off + N && off >= 0, SLAMHIP_E_ARG, "bias_act_slice_fwd: bad shape");
  if (M == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(z && bias && (y_bf16 || y_f32), SLAMHIP_E_ARG, "bias_act_slice_fwd: null pointer");
  const long long total = M * N;
  hipLaunchKernelGGL(bias_act_slice_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, z, bias,
                     (bf16_t*)y_bf16, y_f32, total, N, Np, ld, off, relu);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bias_act_slice_bwd(const void* dy_bf16, const float* dy_f32, const void* y_bf16, void* dz, float* dbias,
                               long long M, int N, int Np, int ld, int off, int relu, void* stream) {
  SLAMHIP_REQUIRE(M >= 0 && N > 0 && Np >= N && ld >= off + N && off >= 0, SLAMHIP_E_ARG, "bias_act_slice_bwd: bad shape");
  SLAMHIP_REQUIRE(dbias && (M == 0 || ((dy_bf16 || dy_f32) && dz && (y_bf16 || !relu))), SLAMHIP_E_ARG,
                  "bias_act_slice_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  SLAMHIP_HIP(zero_async(dbias, sizeof(float) * (size_t)Np, s));
  if (M == 0) return SLAMHIP_OK;
  long long slices = (M + 1023) / 1024;
  if (slices > 1024) slices = 1024;
  const long long rows_per_block = (M + slices - 1) / slices;
  hipLaunchKernelGGL(bias_act_slice_bwd_kernel, dim3((Np + 63) / 64, (unsigned)slices), dim3(256), 0, s,
                     (const bf16_t*)dy_bf16, dy_f32, (const bf16_t*)y_bf16, (bf16_t*)dz, dbias, M, N, Np, ld, off, relu,
                     rows_per_block);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

static int pool_geometry(int B, int H, int W, int C, int K, int stride, int same, int is_max, PoolGeom* g) {
  g->B = B; g->H = H; g->W = W; g->C = C; g->K = K; g->stride = stride; g->is_max = is_max;
  return same_geometry(H, K, stride, same, &g->Ho, &g->pad_t) && same_geometry(W, K, stride, same, &g->Wo, &g->pad_l);
}

int slamhip_pool_fwd(const void* x, void* y, int B, int H, int W, int C, int K, int stride, int same, int is_max,
                     void* stream) {
  PoolGeom g;
  SLAMHIP_REQUIRE(B >= 0 && C > 0 && pool_geometry(B, H, W, C, K, stride, same, is_max, &g), SLAMHIP_E_ARG,
                  "pool_fwd: bad shape");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(x && y, SLAMHIP_E_ARG, "pool_fwd: null pointer");
  const long long total = (long long)B * g.Ho * g.Wo * C;
  hipLaunchKernelGGL(pool_fwd_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, g, total);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_pool_bwd_add(const void* x, const void* dy, void* dx, int B, int H, int W, int C, int K, int stride,
                         int same, int is_max, void* stream) {
  PoolGeom g;
  SLAMHIP_REQUIRE(B >= 0 && C > 0 && pool_geometry(B, H, W, C, K, stride, same, is_max, &g), SLAMHIP_E_ARG,
                  "pool_bwd_add: bad shape");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(x && dy && dx, SLAMHIP_E_ARG, "pool_bwd_add: null pointer");
  const long long total = (long long)B * H * W * C;
  hipLaunchKernelGGL(pool_bwd_add_kernel, dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)dy, (bf16_t*)dx, g, total);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_lrn_fwd(const void* x, void* y, long long P, int C, int radius, float bias, float alpha, float beta,
                    void* stream) {
  SLAMHIP_REQUIRE(P >= 0 && C > 0 && radius >= 0, SLAMHIP_E_ARG, "lrn_fwd: bad shape");
  if (P == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(x && y, SLAMHIP_E_ARG, "lrn_fwd: null pointer");
  hipLaunchKernelGGL(lrn_fwd_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (bf16_t*)y, P * C, C, radius, bias, alpha, beta);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_lrn_bwd_add(const void* x, const void* dy, void* dx, long long P, int C, int radius, float bias, float alpha,
                        float beta, void* stream) {
  SLAMHIP_REQUIRE(P >= 0 && C > 0 && radius >= 0, SLAMHIP_E_ARG, "lrn_bwd_add: bad shape");
  if (P == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(x && dy && dx, SLAMHIP_E_ARG, "lrn_bwd_add: null pointer");
  hipLaunchKernelGGL(lrn_bwd_add_kernel, dim3(grid_for(P * C)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x,
                     (const bf16_t*)dy, (bf16_t*)dx, P * C, C, radius, bias, alpha, beta);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_pad_cast_bf16(const float* src, void* dst, long long P, int C, int Cp, void* stream) {
  SLAMHIP_REQUIRE(P >= 0 && C > 0 && Cp >= C, SLAMHIP_E_ARG, "pad_cast: bad shape");
  if (P == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(src && dst, SLAMHIP_E_ARG, "pad_cast: null pointer");
  hipLaunchKernelGGL(pad_cast_kernel, dim3(grid_for(P * Cp)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                     P * Cp, C, Cp);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

}  // extern "C"
