// Zero fill as an ordinary kernel, for any address and byte count.  The library clears every accumulator with
// this and never with hipMemsetAsync: inside a captured HIP graph a memset becomes a node of its own kind, and
// replays of a training step that held some fifty of them gave wrong sums now and then, while the same step made
// of kernel nodes alone replays correctly.
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

namespace slamhip {
namespace {

// bytes before the first 16-byte boundary and after the last one are written one at a time by the first
// workgroup, the aligned body in 16-byte stores
__global__ void __launch_bounds__(256) zero_bytes_kernel(unsigned char* __restrict__ p, size_t head, size_t n16, size_t tail) {
  uint4* const body = reinterpret_cast<uint4*>(p + head);
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) body[i] = make_uint4(0u, 0u, 0u, 0u);
  if (blockIdx.x == 0) {
    if (threadIdx.x < head) p[threadIdx.x] = 0;
    if (threadIdx.x < tail) p[head + n16 * 16 + threadIdx.x] = 0;
  }
}

inline hipError_t zero_async(void* ptr, size_t bytes, hipStream_t s) {
  if (bytes == 0) return hipSuccess;
#ifdef SLAMHIP_ZERO_BY_MEMSET   // diagnostic build only: the memset nodes this file exists to avoid
  return hipMemsetAsync(ptr, 0, bytes, s);
#endif
  size_t head = (16 - (reinterpret_cast<uintptr_t>(ptr) & 15)) & 15;
  if (head > bytes) head = bytes;
  const size_t n16 = (bytes - head) / 16, tail = bytes - head - n16 * 16;
  size_t blocks = (n16 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(zero_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, s, (unsigned char*)ptr, head, n16, tail);
  return hipGetLastError();
}

}  // namespace
}  // namespace slamhip
