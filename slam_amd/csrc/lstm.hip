// Recurrent head of the CNN-LSTM trainer (slam/network/lstm_model.py:25-49): a stack of
// L LSTM cells of n units applied for ONE time step to the encoder output, then a
// linear output layer.  float32 throughout; the work is a few thousand multiply-adds
// per batch row, so one workgroup per row runs the whole stack out of LDS.
//
// Cell (TensorFlow 0.x LSTMCell without peepholes or projection, the class the
// reference instantiates): with in = [x, m_prev] (2n wide) and W [2n][4n],
//   (i, j, f, o) = split(in . W + b, 4)
//   c = sigmoid(f + forget_bias) * c_prev + sigmoid(i) * tanh(j)
//   m = sigmoid(o) * tanh(c)
// The state of a layer is [c, m] (2n wide); the stack's state is the layers' states
// side by side (MultiRNNCell), the input of layer l > 0 is m of layer l - 1.
#include <hip/hip_runtime.h>

#include "common.h"

namespace slamhip {

__device__ __forceinline__ float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

// gates [B][L][4n]: the activated (i, j, f, o); state_out [B][L][2n]; out [B][D]
__global__ void __launch_bounds__(256)
lstm_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ state_in, const float* __restrict__ W,
                     const float* __restrict__ bias, const float* __restrict__ fcw, const float* __restrict__ fcb,
                     float forget_bias, int n, int L, int D, float* __restrict__ gates, float* __restrict__ state_out,
                     float* __restrict__ out) {
  extern __shared__ float sm[];
  float* in = sm;            // 2n
  float* act = sm + 2 * n;   // 4n
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < n) in[t] = x[(long long)b * n + t];
  for (int l = 0; l < L; ++l) {
    const long long row = (long long)b * L + l;
    if (t < n) in[n + t] = state_in[row * 2 * n + n + t];
    __syncthreads();
    if (t < 4 * n) {
      float s = bias[l * 4 * n + t];
      const float* w = W + (long long)l * 2 * n * 4 * n + t;
      for (int k = 0; k < 2 * n; ++k) s = fmaf(in[k], w[(long long)k * 4 * n], s);
      const int q = t / n;
      const float a = q == 1 ? tanhf(s) : sigmoidf_(q == 2 ? s + forget_bias : s);
      act[t] = a;
      gates[row * 4 * n + t] = a;
    }
    __syncthreads();
    if (t < n) {
      const float c = act[2 * n + t] * state_in[row * 2 * n + t] + act[t] * act[n + t];
      const float m = act[3 * n + t] * tanhf(c);
      state_out[row * 2 * n + t] = c;
      state_out[row * 2 * n + n + t] = m;
      in[t] = m;
    }
    __syncthreads();
  }
  if (t < D) {
    float s = fcb[t];
    for (int k = 0; k < n; ++k) s = fmaf(in[k], fcw[k * D + t], s);
    out[(long long)b * D + t] = s;
  }
}

// d_pre [B][L][4n]: gradient at the gate pre-activations; d_x [B][n]
__global__ void __launch_bounds__(256)
lstm_head_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ state_in, const float* __restrict__ W,
                     const float* __restrict__ fcw, const float* __restrict__ gates,
                     const float* __restrict__ state_out, int n, int L, int D, float* __restrict__ d_pre,
                     float* __restrict__ d_x) {
  extern __shared__ float sm[];
  float* dm = sm;           // n
  float* dp = sm + n;       // 4n
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < n) {
    float s = 0.0f;
    for (int d = 0; d < D; ++d) s = fmaf(d_out[(long long)b * D + d], fcw[t * D + d], s);
    dm[t] = s;
  }
  __syncthreads();
  for (int l = L - 1; l >= 0; --l) {
    const long long row = (long long)b * L + l;
    if (t < n) {
      const float* g = gates + row * 4 * n;
      const float gi = g[t], gj = g[n + t], gf = g[2 * n + t], go = g[3 * n + t];
      const float tc = tanhf(state_out[row * 2 * n + t]);
      const float c_prev = state_in[row * 2 * n + t];
      const float d_m = dm[t];
      const float dc = d_m * go * (1.0f - tc * tc);
      dp[t] = dc * gj * gi * (1.0f - gi);
      dp[n + t] = dc * gi * (1.0f - gj * gj);
      dp[2 * n + t] = dc * c_prev * gf * (1.0f - gf);
      dp[3 * n + t] = d_m * tc * go * (1.0f - go);
    }
    __syncthreads();
    if (t < 4 * n) d_pre[row * 4 * n + t] = dp[t];
    float s = 0.0f;
    if (t < n) {   // gradient at this layer's input (rows 0..n-1 of W); the m_prev half is a fed constant
      const float* w = W + ((long long)l * 2 * n + t) * 4 * n;
      for (int g = 0; g < 4 * n; ++g) s = fmaf(w[g], dp[g], s);
    }
    __syncthreads();
    if (t < n) dm[t] = s;
    __syncthreads();
  }
  if (t < n) d_x[(long long)b * n + t] = dm[t];
}

// one thread per parameter: sum over the batch in row order (deterministic)
__global__ void __launch_bounds__(256)
lstm_head_param_grad_kernel(const float* __restrict__ x, const float* __restrict__ state_in,
                            const float* __restrict__ state_out, const float* __restrict__ d_pre,
                            const float* __restrict__ d_out, int B, int n, int L, int D, float* __restrict__ dW,
                            float* __restrict__ dbias, float* __restrict__ dfcw, float* __restrict__ dfcb) {
  const int nW = L * 2 * n * 4 * n, nb = L * 4 * n, nF = n * D;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nW) {
    const int g = i % (4 * n), k = (i / (4 * n)) % (2 * n), l = i / (8 * n * n);
    float s = 0.0f;
    for (int b = 0; b < B; ++b) {
      const long long row = (long long)b * L + l;
      float v;
      if (k >= n) v = state_in[row * 2 * n + k];                       // m_prev
      else if (l == 0) v = x[(long long)b * n + k];
      else v = state_out[(row - 1) * 2 * n + n + k];                   // m of the layer below
      s = fmaf(v, d_pre[row * 4 * n + g], s);
    }
    dW[i] = s;
    return;
  }
  i -= nW;
  if (i < nb) {
    const int g = i % (4 * n), l = i / (4 * n);
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += d_pre[((long long)b * L + l) * 4 * n + g];
    dbias[i] = s;
    return;
  }
  i -= nb;
  if (i < nF) {
    const int d = i % D, k = i / D;
    float s = 0.0f;
    for (int b = 0; b < B; ++b)
      s = fmaf(state_out[((long long)b * L + L - 1) * 2 * n + n + k], d_out[(long long)b * D + d], s);
    dfcw[i] = s;
    return;
  }
  i -= nF;
  if (i < D) {
    float s = 0.0f;
    for (int b = 0; b < B; ++b) s += d_out[(long long)b * D + i];
    dfcb[i] = s;
  }
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

int slamhip_lstm_head_fwd(const float* x, const float* state_in, const float* W, const float* bias, const float* fc_w,
                          const float* fc_b, float forget_bias, int B, int n, int L, int D, float* gates,
                          float* state_out, float* out, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && n > 0 && n <= 64 && L > 0 && D > 0 && D <= 64, SLAMHIP_E_ARG,
                  "lstm_head_fwd: need 1 <= units <= 64, layers >= 1, 1 <= outputs <= 64");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(x && state_in && W && bias && fc_w && fc_b && gates && state_out && out, SLAMHIP_E_ARG,
                  "lstm_head_fwd: null pointer");
  const int threads = ((4 * n > D ? 4 * n : D) + 63) / 64 * 64;
  hipLaunchKernelGGL(lstm_head_fwd_kernel, dim3(B), dim3(threads), 6 * n * sizeof(float), (hipStream_t)stream, x,
                     state_in, W, bias, fc_w, fc_b, forget_bias, n, L, D, gates, state_out, out);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_lstm_head_bwd(const float* d_out, const float* x, const float* state_in, const float* W, const float* fc_w,
                          const float* gates, const float* state_out, int B, int n, int L, int D, float* d_pre,
                          float* d_x, float* dW, float* dbias, float* dfc_w, float* dfc_b, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && n > 0 && n <= 64 && L > 0 && D > 0 && D <= 64, SLAMHIP_E_ARG,
                  "lstm_head_bwd: need 1 <= units <= 64, layers >= 1, 1 <= outputs <= 64");
  SLAMHIP_REQUIRE(dW && dbias && dfc_w && dfc_b, SLAMHIP_E_ARG, "lstm_head_bwd: null gradient pointer");
  SLAMHIP_REQUIRE(B == 0 || (d_out && x && state_in && W && fc_w && gates && state_out && d_pre && d_x), SLAMHIP_E_ARG,
                  "lstm_head_bwd: null pointer");
  hipStream_t s = (hipStream_t)stream;
  if (B > 0) {
    const int threads = (4 * n + 63) / 64 * 64;
    hipLaunchKernelGGL(lstm_head_bwd_kernel, dim3(B), dim3(threads), 5 * n * sizeof(float), s, d_out, state_in, W, fc_w,
                       gates, state_out, n, L, D, d_pre, d_x);
  }
  const int total = L * 2 * n * 4 * n + L * 4 * n + n * D + D;
  hipLaunchKernelGGL(lstm_head_param_grad_kernel, dim3((total + 255) / 256), dim3(256), 0, s, x, state_in, state_out,
                     d_pre, d_out, B, n, L, D, dW, dbias, dfc_w, dfc_b);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

}  // extern "C"
