// Calibration of rocprofv3's FETCH_SIZE on the access widths of the fused frame-pair kernel: each kernel
// below reads every byte of a buffer of known size exactly once with one of that kernel's load shapes
// (image geometry 640x480 like the kernel's), so  known bytes / (FETCH_SIZE * 1024)  is the factor by which
// the counter under-reports that shape.  Run under  rocprofv3 --pmc FETCH_SIZE -- build/fetch_calibration
// and read the per-kernel counter values (tools/pmc_traffic.sh does both and writes the factors as JSON).
//   cal_u16_lane      2 bytes per lane, 64 consecutive columns per wave, 16 rows per lane   (anchor depth)
//   cal_q8_stride3    8 bytes per lane at 3-byte pixel stride, pulled back to 4-byte alignment (anchor colour)
//   cal_x16_chunk     16 bytes per thread, 8-pixel chunks of a 160-pixel window row          (window depth)
//   cal_q8x3_chunk    three 8-byte loads per thread, 24-byte chunks of a window row          (window colour)
//   cal_mix           all four in the proportions of one tile of the kernel (2 : 3 : 2 : 3 bytes)
// build: hipcc --offload-arch=gfx950 -O3 -o build/fetch_calibration tools/microbench/fetch_calibration.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 640, H = 480, PLANES = 192;
struct Dword2 { unsigned int x, y; };

__device__ __forceinline__ void sink(unsigned int v, unsigned int* out) {
  if (v == 0x79b9u) out[threadIdx.x] = v;   // never true for the zero-filled buffers (but possible for 16-bit data); keeps the loads alive
}

// block = 128 threads = two waves side by side: 128 columns x 16 rows of one plane
__global__ __launch_bounds__(128) void cal_u16_lane(const uint16_t* __restrict__ d, unsigned int* out) {
  const int col = blockIdx.x * 128 + threadIdx.x, row0 = blockIdx.y * 16;
  const uint16_t* p = d + (size_t)blockIdx.z * W * H;
  unsigned int v[16], acc = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = p[(row0 + r) * W + col];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc ^= v[r];
  sink(acc, out);
}

__global__ __launch_bounds__(128) void cal_q8_stride3(const uint8_t* __restrict__ c, unsigned int* out) {
  const int col = blockIdx.x * 128 + threadIdx.x, row0 = blockIdx.y * 16;
  const uint8_t* p = c + (size_t)blockIdx.z * W * H * 3;
  const unsigned int L = 3u * col, La = min(L & ~3u, 3u * W - 8u);
  Dword2 v[16];
  unsigned int acc = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = *reinterpret_cast<const Dword2*>(p + (size_t)(row0 + r) * 3 * W + La);
#pragma unroll
  for (int r = 0; r < 16; ++r) acc ^= v[r].x ^ v[r].y;
  sink(acc, out);
}

// block = 256 threads: a window of 160 x 24 pixels = 20 x 24 chunks of 8 pixels, two chunks per thread at most
__global__ __launch_bounds__(256) void cal_x16_chunk(const uint16_t* __restrict__ d, unsigned int* out) {
  const int wx0 = blockIdx.x * 160, wy0 = blockIdx.y * 24;
  const uint16_t* p = d + (size_t)blockIdx.z * W * H;
  uint4 v[2];
  unsigned int acc = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int c = threadIdx.x + q * 256;
    if (c >= 480) c = 0;
    const int cr = c / 20, cc = c - cr * 20;
    v[q] = *reinterpret_cast<const uint4*>(p + (wy0 + cr) * W + wx0 + cc * 8);
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
  sink(acc, out);
}

__global__ __launch_bounds__(256) void cal_q8x3_chunk(const uint8_t* __restrict__ c3, unsigned int* out) {
  const int wx0 = blockIdx.x * 160, wy0 = blockIdx.y * 24;
  const uint8_t* p = c3 + (size_t)blockIdx.z * W * H * 3;
  Dword2 v[2][3];
  unsigned int acc = 0;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int c = threadIdx.x + q * 256;
    if (c >= 480) c = 0;
    const int cr = c / 20, cc = c - cr * 20;
    const Dword2* s = reinterpret_cast<const Dword2*>(p + 3 * ((size_t)(wy0 + cr) * W + wx0 + cc * 8));
    v[q][0] = s[0]; v[q][1] = s[1]; v[q][2] = s[2];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 3; ++i) acc ^= v[q][i].x ^ v[q][i].y;
  sink(acc, out);
}

// one tile of the fused kernel: 128 x 32 anchor pixels (four waves, 16 rows per lane) of frame 1 and a
// window of 128 x 32 pixels of frame 2 (16 x 32 chunks, two per thread)
__global__ __launch_bounds__(256) void cal_mix(const uint16_t* __restrict__ d1, const uint8_t* __restrict__ c1,
                                               const uint16_t* __restrict__ d2, const uint8_t* __restrict__ c2, unsigned int* out) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int col = blockIdx.x * 128 + (wave & 1) * 64 + lane, row0 = blockIdx.y * 32 + (wave >> 1) * 16;
  const size_t plane = (size_t)blockIdx.z * W * H;
  const unsigned int L = 3u * col, La = min(L & ~3u, 3u * W - 8u);
  unsigned int a_d[16];
  Dword2 a_q[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    a_d[r] = d1[plane + (row0 + r) * W + col];
    a_q[r] = *reinterpret_cast<const Dword2*>(c1 + 3 * plane + (size_t)(row0 + r) * 3 * W + La);
  }
  uint4 v[2];
  Dword2 q3[2][3];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int c = threadIdx.x + q * 256, cr = c >> 4, cc = c & 15;
    const size_t o = plane + (size_t)(blockIdx.y * 32 + cr) * W + blockIdx.x * 128 + cc * 8;
    v[q] = *reinterpret_cast<const uint4*>(d2 + o);
    const Dword2* s = reinterpret_cast<const Dword2*>(c2 + 3 * o);
    q3[q][0] = s[0]; q3[q][1] = s[1]; q3[q][2] = s[2];
  }
  unsigned int acc = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc ^= a_d[r] ^ a_q[r].x ^ a_q[r].y;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    acc ^= v[q].x ^ v[q].y ^ v[q].z ^ v[q].w;
#pragma unroll
    for (int i = 0; i < 3; ++i) acc ^= q3[q][i].x ^ q3[q][i].y;
  }
  sink(acc, out);
}

int main() {
  const size_t px = (size_t)PLANES * W * H;
  uint16_t *d1, *d2; uint8_t *c1, *c2; unsigned int* out;
  CHECK(hipMalloc(&d1, 2 * px)); CHECK(hipMalloc(&d2, 2 * px));
  CHECK(hipMalloc(&c1, 3 * px + 64)); CHECK(hipMalloc(&c2, 3 * px + 64));
  CHECK(hipMalloc(&out, 4096));
  CHECK(hipMemset(d1, 0, 2 * px)); CHECK(hipMemset(d2, 0, 2 * px));
  CHECK(hipMemset(c1, 0, 3 * px + 64)); CHECK(hipMemset(c2, 0, 3 * px + 64));
  CHECK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    // the buffers alternate so that a kernel never finds its lines left in the caches by the one before
    cal_u16_lane<<<dim3(W / 128, H / 16, PLANES), 128>>>(rep & 1 ? d2 : d1, out);
    cal_q8_stride3<<<dim3(W / 128, H / 16, PLANES), 128>>>(rep & 1 ? c2 : c1, out);
    cal_x16_chunk<<<dim3(W / 160, H / 24, PLANES), 256>>>(rep & 1 ? d1 : d2, out);
    cal_q8x3_chunk<<<dim3(W / 160, H / 24, PLANES), 256>>>(rep & 1 ? c1 : c2, out);
    cal_mix<<<dim3(W / 128, H / 32, PLANES), 256>>>(d1, c1, d2, c2, out);
    CHECK(hipDeviceSynchronize());
  }
  CHECK(hipGetLastError());
  printf("{\"planes\": %d, \"bytes\": {\"cal_u16_lane\": %zu, \"cal_q8_stride3\": %zu, \"cal_x16_chunk\": %zu, \"cal_q8x3_chunk\": %zu, \"cal_mix\": %zu}}\n",
         PLANES, 2 * px, 3 * px, 2 * px, 3 * px, 10 * px);
  return 0;
}
