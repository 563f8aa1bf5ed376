// Issue cost (cycles per wave64 instruction per SIMD, 4 waves/SIMD resident) of
// the individual VALU operations the frame-pair kernel is made of, on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e_)); return; } } while (0)

typedef _Float16 h2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) op_kernel(float* out, unsigned long long* clk, int iters) {
  const int tid = threadIdx.x + blockIdx.x * blockDim.x;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  double d[8];
  float f[8];
  unsigned int u[8];
  for (int i = 0; i < 8; ++i) { d[i] = tid * 1e-3 + i + 0.25; f[i] = tid * 1e-3f + i + 0.25f; u[i] = tid * 7 + i; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (KIND == 0) { asm volatile("v_cvt_f64_u32 %0, %1" : "=v"(d[i]) : "v"(u[i])); }
      else if (KIND == 1) { asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[i]) : "v"(d[i])); }
      else if (KIND == 2) { asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(d[i])); }
      else if (KIND == 3) { asm volatile("v_fract_f64 %0, %1" : "=v"(d[i]) : "v"(d[i])); }
      else if (KIND == 4) { asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(d[i]) : "v"(f[i])); }
      else if (KIND == 5) { asm volatile("v_floor_f64 %0, %1" : "=v"(d[i]) : "v"(d[i])); }
      else if (KIND == 6) { asm volatile("v_add_f64 %0, %1, %1" : "=v"(d[i]) : "v"(d[i])); }
      else if (KIND == 7) { asm volatile("v_mul_f64 %0, %1, %1" : "=v"(d[i]) : "v"(d[i])); }
      else if (KIND == 8) { asm volatile("v_cmp_le_f64 vcc, %0, %0\n\tv_cndmask_b32 %1, 0, %1, vcc" : : "v"(d[i]), "v"(u[i]) : "vcc"); }
      else if (KIND == 9) { asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(u[i]) : : ); }
      else if (KIND == 10) { asm volatile("v_perm_b32 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
      else if (KIND == 11) { asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
      else if (KIND == 12) { asm volatile("v_fma_mix_f32 %0, %1, %2, %2 op_sel_hi:[0,1,1]" : "=v"(f[i]) : "v"(f[i]), "v"(u[i])); }
      else if (KIND == 13) { asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(f[i]) : "v"(u[i])); }
      else if (KIND == 14) { asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
      else if (KIND == 15) { asm volatile("v_cvt_f32_u32_sdwa %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "=v"(f[i]) : "v"(u[i])); }
      else if (KIND == 16) { asm volatile("v_add_f32 %0, |%0|, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7])); }
      else if (KIND == 17) { asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(d[i]) : "v"(d[(i + 1) & 7])); }
      else if (KIND == 18) { asm volatile("v_rcp_f64 %0, %1" : "=v"(d[i]) : "v"(d[i])); }
      else if (KIND == 19) { asm volatile("v_min3_u32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7])); }
      else if (KIND == 20) { asm volatile("v_cmp_lt_u32 vcc, %0, %1" : : "v"(u[i]), "v"(u[(i + 1) & 7]) : "vcc"); }
      else if (KIND == 21) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(d[i]) : "v"(u[i]) : "vcc"); }
    }
  }
  float acc = 0.f;
  for (int i = 0; i < 8; ++i) acc += (float)d[i] + f[i] + (float)u[i];
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[tid] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int instr_per_op) {
  const int iters = 4000, blocks = 256 * 4;
  float* out; unsigned long long* clk;
  CHECK(hipMalloc(&out, sizeof(float) * 256 * blocks));
  CHECK(hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(op_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  CHECK(hipEventRecord(e0));
  const int reps = 10;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(op_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
  CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * blocks);
  CHECK(hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost));
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double winstr = (double)blocks * 4 * (double)iters * 8 * instr_per_op * reps;
  const double per_simd_per_s = winstr / (ms * 1e-3) / 1024.0;
  printf("%-28s %.2f cycles/instr (clock %.2f GHz)\n", name, ghz * 1e9 / per_simd_per_s, ghz);
  CHECK(hipFree(out)); CHECK(hipFree(clk));
}

int main() {
  run<0>("v_cvt_f64_u32", 1); run<1>("v_cvt_i32_f64", 1); run<2>("v_cvt_f32_f64", 1); run<3>("v_fract_f64", 1);
  run<4>("v_cvt_f64_f32", 1); run<5>("v_floor_f64", 1); run<6>("v_add_f64", 1); run<7>("v_mul_f64", 1);
  run<8>("v_cmp_le_f64+v_cndmask", 2); run<9>("v_cndmask_b32", 1); run<10>("v_perm_b32", 1); run<11>("v_pk_add_f16", 1);
  run<12>("v_fma_mix_f32", 1); run<13>("v_cvt_f32_ubyte0", 1); run<14>("v_mov_b32_dpp", 1); run<15>("v_cvt_f32_u32_sdwa", 1);
  run<16>("v_add_f32 |abs|", 1); run<17>("v_fma_f64", 1); run<18>("v_rcp_f64", 1); run<19>("v_min3_u32", 1);
  run<20>("v_cmp_lt_u32", 1); run<21>("v_mad_u64_u32", 1);
  return 0;
}
