// VALU issue-rate microbenchmark for gfx950: how many wave64 instructions per
// second one MI355X retires for f32 fma, packed f32 fma, f64 fma and f64 rcp,
// plus the shader clock seen from inside the kernel (s_memtime vs s_memrealtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(256) rate_kernel(float* out, unsigned long long* clk, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  float acc = 0.f;
  if (KIND == 0) {  // f32 fma, 8 independent chains
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = tid * 1e-9f + i;
    const float m = 1.0000001f, c = 1e-7f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], m, c);
    }
    for (int i = 0; i < 8; ++i) acc += a[i];
  } else if (KIND == 1) {  // packed f32 fma, 8 chains of float2
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 a[8];
    for (int i = 0; i < 8; ++i) a[i] = f2{tid * 1e-9f + i, tid * 1e-9f - i};
    const f2 m = {1.0000001f, 0.9999999f}, c = {1e-7f, 1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
    }
    for (int i = 0; i < 8; ++i) acc += a[i].x + a[i].y;
  } else if (KIND == 2) {  // f64 fma
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = tid * 1e-9 + i;
    const double m = 1.0000001, c = 1e-7;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_fma(a[i], m, c);
    }
    for (int i = 0; i < 8; ++i) acc += (float)a[i];
  } else if (KIND == 3) {  // f64 rcp
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = tid * 1e-9 + i + 1.5;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_rcp(a[i]);
    }
    for (int i = 0; i < 8; ++i) acc += (float)a[i];
  } else if (KIND == 4) {  // f64 compare + select (v_cmp_le_f64 + v_cndmask)
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = tid * 1e-9 + i;
    int cnt = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { cnt += (a[i] <= (double)it) ? 1 : 0; }
    }
    acc = (float)cnt;
  } else if (KIND == 5) {  // v_cvt_f32_ubyte0 + add
    unsigned int a[8];
    for (int i = 0; i < 8; ++i) a[i] = tid + i;
    float s[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { s[i] += (float)((a[i] + it) & 0xffu); }
    }
    for (int i = 0; i < 8; ++i) acc += s[i];
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = acc;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

template <int KIND>
void run(const char* name, int per_iter_instrs, int waves_per_simd) {
  const int iters = 20000;
  const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  float* out;
  unsigned long long* clk;
  hipMalloc(&out, sizeof(float) * 256 * blocks);
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e0);
  const int reps = 20;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(rate_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double winstr = (double)blocks * 4 * (double)iters * per_iter_instrs * reps;
  const double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
  printf("%-22s waves/SIMD=%d  %.3f ms/launch  %.2f G wave-instr/s/SIMD  clock %.2f GHz  => %.2f cycles/wave-instr\n",
         name, waves_per_simd, ms / reps, per_simd_per_s * 1e-9, ghz, ghz * 1e9 / per_simd_per_s);
  hipFree(out);
  hipFree(clk);
}

int main() {
  for (int w : {1, 2, 4}) {
    if (w == 1) {
      run<0>("v_fma_f32", 8, 1); run<1>("v_pk_fma_f32", 8, 1); run<2>("v_fma_f64", 8, 1); run<3>("v_rcp_f64", 8, 1);
    } else if (w == 2) {
      run<0>("v_fma_f32", 8, 2); run<1>("v_pk_fma_f32", 8, 2); run<2>("v_fma_f64", 8, 2); run<3>("v_rcp_f64", 8, 2);
    } else {
      run<0>("v_fma_f32", 8, 4); run<1>("v_pk_fma_f32", 8, 4); run<2>("v_fma_f64", 8, 4); run<3>("v_rcp_f64", 8, 4);
      run<4>("v_cmp_f64+addc", 16, 4); run<5>("add+and+cvt_ubyte+add", 32, 4);
    }
  }
  return 0;
}
