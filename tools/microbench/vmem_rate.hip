// Issue rate of vector-memory loads of different widths on gfx950 with the data resident in L2 / L1:
// cycles per wave64 load instruction per CU (all CUs busy, 12 waves per CU), and the bytes per cycle it gives.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int KIND>
__global__ void __launch_bounds__(256) load_kernel(const unsigned char* __restrict__ src, unsigned int* out,
                                                   unsigned long long* clk, int iters, unsigned int span) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  unsigned int acc = 0;
  // every wave walks its own window of `span` bytes (L2-resident), rows of 64 lanes x width bytes
  unsigned int off = ((blockIdx.x * 4 + wave) * 4096u) % span;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const unsigned int o = (off + u * 1280u) % span;
      if (KIND == 0) acc += *reinterpret_cast<const unsigned short*>(src + o + 2 * lane);
      else if (KIND == 1) acc += *reinterpret_cast<const unsigned int*>(src + o + 4 * lane);
      else if (KIND == 2) { const uint2 v = *reinterpret_cast<const uint2*>(src + o + 8 * lane); acc += v.x ^ v.y; }
      else if (KIND == 3) { const uint4 v = *reinterpret_cast<const uint4*>(src + o + 16 * lane); acc += v.x ^ v.y ^ v.z ^ v.w; }
      else if (KIND == 4) { struct __attribute__((aligned(4))) D2 { unsigned x, y; }; const D2 v = *reinterpret_cast<const D2*>(src + o + ((6 * lane) & ~3)); acc += v.x ^ v.y; }
      else if (KIND == 5) acc += src[o + 3 * lane];
    }
    off = (off + 10240u) % span;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * 256 + threadIdx.x] = acc;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = t1 - t0; clk[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int bytes_per_lane) {
  const int iters = 2000, blocks = 256 * 3;
  const unsigned int span = 1u << 20;   // 1 MiB: L2 resident
  unsigned char* src; unsigned int* out; unsigned long long* clk;
  hipMalloc(&src, span + 65536); hipMemset(src, 1, span + 65536);
  hipMalloc(&out, sizeof(unsigned int) * 256 * blocks);
  hipMalloc(&clk, sizeof(unsigned long long) * 2 * blocks);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(load_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters, span);
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(load_kernel<KIND>, dim3(blocks), dim3(256), 0, 0, src, out, clk, iters, span);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(2 * blocks);
  hipMemcpy(h.data(), clk, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
  const double ghz = (double)h[0] / (double)h[1] * 0.1;
  const double winstr_per_cu = (double)blocks / 256.0 * 4 * (double)iters * 8 * reps;
  const double cyc = ms * 1e-3 * ghz * 1e9 / winstr_per_cu;
  printf("%-28s %.1f cycles per wave-instruction per CU, %.1f B/cycle/CU (clock %.2f GHz)\n", name, cyc, 64.0 * bytes_per_lane / cyc, ghz);
  hipFree(src); hipFree(out); hipFree(clk);
}

int main() {
  run<0>("global_load_ushort", 2); run<1>("global_load_dword", 4); run<2>("global_load_dwordx2", 8);
  run<3>("global_load_dwordx4", 16); run<4>("dwordx2 at 6-byte lane stride", 6); run<5>("global_load_ubyte stride 3", 1);
  return 0;
}
