// Pose-regression encoder kernels for gfx950 (MI355X): the VGG16-style network
// of slam/network/cnn_model.py:31-80 (conv3x3 SAME + batch-norm with batch
// statistics + ReLU, 2x2 max-pool, two ReLU fully connected layers, linear
// output), its weighted L2 loss (cnn_model.py:139-140) and Adam
// (cnn_model.py:162-166, TensorFlow defaults).
//
// Layout: activations NHWC bf16; master parameters float32 in the reference's
// layouts (conv HWIO [3,3,Cin,Cout], fc [in,out]); bf16 operand copies are made
// per step in the K-contiguous layouts the MFMA kernels read.
// The dense contractions (conv forward, data gradient, fully connected) are one
// implicit-GEMM kernel on v_mfma_f32_32x32x16_bf16 with float32 accumulation;
// the weight gradient is a second MFMA kernel that reads its pixel-major LDS
// tiles through the transposing ds_read_b64_tr_b16.  Everything else is
// bandwidth-bound elementwise / per-channel reduction code.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <mutex>

#include "common.h"
#include "zero_fill.h"

namespace slamhip {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short short4_t __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int LDS_PITCH = 72;   // 64 bf16 of payload + 8 of padding per row (144 B)

// ---------------------------------------------------------------------------
// How partial results meet.  By default the kernels that split a sum over workgroups (channel statistics, the
// batch-norm backward sums, filter-gradient slices, split-K products) add their parts with float atomics into a
// cleared output: fast, and the order of the additions - hence the last bits of the sums - differs from run to run.
// SLAMHIP_DETERMINISTIC=1 (read once per process) selects fixed-order sums instead: every part is stored to its own
// slot of a scratch buffer and ordered_sum_kernel adds the slots in index order, so two runs on the same inputs give
// the same bits.  The scratch buffer is the one piece of device memory the library owns: one grow-only allocation
// per device, made outside stream capture, shared by consecutive calls on one stream (calls of this mode on
// different streams of one device would race on it).
// ---------------------------------------------------------------------------
static bool deterministic_mode() {
  static const bool on = [] { const char* e = getenv("SLAMHIP_DETERMINISTIC"); return e && atoi(e) != 0; }();
  return on;
}

static float* det_scratch(size_t bytes) {
  static std::mutex mu;
  static void* buf[64] = {};
  static size_t cap[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (cap[dev] < bytes) {
    if (buf[dev]) (void)hipFree(buf[dev]);     // waits for the device: no kernel still reads the old buffer
    buf[dev] = nullptr;
    cap[dev] = 0;
    const size_t want = (bytes + (size_t(1) << 20)) & ~((size_t(1) << 20) - 1);
    if (hipMalloc(&buf[dev], want) != hipSuccess) return nullptr;
    cap[dev] = want;
  }
  return (float*)buf[dev];
}

// out[i] = ((part[0][i] + part[1][i]) + part[2][i]) + ... over the S slots, in that order
__global__ void __launch_bounds__(256)
ordered_sum_kernel(const float* __restrict__ part, float* __restrict__ out, int S, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  int sl = 0;
  for (; sl + 7 < S; sl += 8) {        // eight loads in flight, added in slot order
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(long long)(sl + j) * n + i];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j];
  }
  for (; sl < S; ++sl) acc += part[(long long)sl * n + i];
  out[i] = acc;
}

static hipError_t ordered_sum(const float* part, float* out, long long S, long long n, hipStream_t s) {
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(ordered_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, out, (int)S, n);
  return hipGetLastError();
}

// A caller that runs many of these entry points in a row (a training step) clears all their accumulators - the
// statistics, the batch-norm backward sums, the filter gradients - with one launch of its own and says so per
// thread (slamhip_caller_clears_accumulators): the entry points then skip their own clearing launches.
static thread_local int t_caller_clears = 0;

static hipError_t clear_accumulator(void* ptr, size_t bytes, hipStream_t s) {
  if (t_caller_clears || deterministic_mode()) return hipSuccess;     // cleared by the caller / overwritten by ordered_sum
  return zero_async(ptr, bytes, s);
}

// Per-column sum and sum of squares of one wave's 32-column slice of an output tile, from the
// values as they are stored (rounded to bfloat16): lanes l and l + 32 hold the same column,
// lanes 0..31 write {sum, sum of squares} of column n to part[n] and part[N + n].
__device__ __forceinline__ void store_column_stats(float s, float q, float* __restrict__ part, int n, int N, int lane) {
  s += __shfl_xor(s, 32);
  q += __shfl_xor(q, 32);
  if (lane < 32 && n < N) { part[n] = s; part[N + n] = q; }
}

// Zeroing by mask keeps the value in registers (a select between two 16-byte values can
// be lowered to an indexed read of a stack copy).
__device__ __forceinline__ uint4 and4(uint4 v, uint32_t keep) {
  return make_uint4(v.x & keep, v.y & keep, v.z & keep, v.w & keep);
}

// ---------------------------------------------------------------------------
// implicit GEMM:  out[m][n] = sum_{tap,c} A[pixel(m) + tap][c] * Bw[n][tap*C + c]
// TAPS = 9: 3x3 SAME convolution over NHWC A (flip = 1 walks the taps backwards:
//           the data gradient with Bw = the flipped, in/out-swapped filter)
// TAPS = 1: plain NT GEMM, A [M][C], Bw [N][C]
// ---------------------------------------------------------------------------
template <int TAPS, int BN, bool WRITE_BF16, bool WRITE_F32>
__global__ void __launch_bounds__(256, BN == 128 ? 3 : 4)   // three 128-wide or four 64-wide workgroups per compute unit
igemm_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Bw, bf16* __restrict__ out_bf16,
                  float* __restrict__ out_f32, int M, int N, int C, int H, int W, int flip, int n_tiles,
                  int total_tiles, float* __restrict__ stat_part, int splitk, int m_tile0) {
  constexpr int BM = 128;
  constexpr int NT = BN / 64;                 // 32-wide MFMA tiles per wave along n
  constexpr int NB = BN / 32;                 // B loader passes: 8 threads (16 B each) cover one 64-element row
  __shared__ __attribute__((aligned(16))) bf16 smem[(BM + BN) * LDS_PITCH];   // A tile, B tile; the output tile at the end
  bf16* const As = smem;
  bf16* const Bs = smem + BM * LDS_PITCH;

  // Workgroups are dealt round-robin to the 8 XCDs; give each XCD one contiguous run of
  // tiles so the n-tiles of a pixel tile and its neighbours (tap halos) share an L2.
  // splitk > 1 (plain GEMM with few tiles): blockIdx.y owns a slice of the K-steps and adds
  // its partial tile to the zeroed float32 output
  const int per_xcd = (total_tiles + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_tiles) return;
  const int m0 = (m_tile0 + work / n_tiles) * BM, n0 = (work % n_tiles) * BN;   // m_tile0: first pixel tile of this launch

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int Ktot = TAPS * C;

  // A loader: thread -> row (pixel) tid>>1, 64-byte half tid&1.  Rows beyond M and taps
  // outside the image load a valid dummy address and are zeroed on the way into LDS.
  const int a_row = tid >> 1, a_half = tid & 1;
  const bool row_ok = m0 + a_row < M;
  const int pm = row_ok ? m0 + a_row : M - 1;
  int py = 0, px = 0;
  if (TAPS == 9) {
    const int hw = H * W;
    const int rem = pm % hw;
    py = rem / W;
    px = rem - py * W;
  }
  const bf16* a_base = A + (long long)pm * C + a_half * 32;
  // B loader: row tid>>3 (+32 per pass), 16-byte chunk tid&7
  const int b_row = tid >> 3, b_chunk = tid & 7;
  const bf16* b_base[NB];
  bool b_ok[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int n = n0 + b_row + i * 32;
    b_ok[i] = n < N;
    b_base[i] = Bw + (long long)(b_ok[i] ? n : N - 1) * Ktot + b_chunk * 8;
  }

  f32x16 acc[2][NT];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps_all = TAPS * (C / 64);
  const int kchunk = (ksteps_all + splitk - 1) / splitk;
  const int ks_begin = (TAPS == 1) ? (int)blockIdx.y * kchunk : 0;
  const int ksteps = (TAPS == 1) ? min(ksteps_all, ks_begin + kchunk) - ks_begin : ksteps_all;
  if (ksteps <= 0) return;
  uint4 ra0, ra1, ra2, ra3, rb[NB];
  bool a_ok;
  int tap = 0, c0 = ks_begin * 64;   // position of the loads in flight

#define SLAMHIP_IGEMM_ISSUE()                                                                   \
  {                                                                                             \
    long long delta = c0;                                                                       \
    a_ok = row_ok;                                                                              \
    if (TAPS == 9) {                                                                            \
      const int t = flip ? 8 - tap : tap;                                                       \
      const int dy = t / 3 - 1, dx = t - (t / 3) * 3 - 1;                                       \
      a_ok = row_ok && (unsigned)(py + dy) < (unsigned)H && (unsigned)(px + dx) < (unsigned)W;  \
      if (a_ok) delta += (long long)(dy * W + dx) * C;                                          \
    }                                                                                           \
    const uint4* src = reinterpret_cast<const uint4*>(a_base + delta);                          \
    ra0 = src[0]; ra1 = src[1]; ra2 = src[2]; ra3 = src[3];                                     \
    _Pragma("unroll") for (int i = 0; i < NB; ++i)                                              \
      rb[i] = *reinterpret_cast<const uint4*>(b_base[i] + tap * C + c0);                        \
    c0 += 64;                                                                                   \
    if (c0 == C) { c0 = 0; ++tap; }                                                             \
  }

  SLAMHIP_IGEMM_ISSUE();
  for (int s = 0; s < ksteps; ++s) {
    __syncthreads();   // the previous step's fragments have been read
    {
      uint4* dst = reinterpret_cast<uint4*>(&As[a_row * LDS_PITCH + a_half * 32]);
      const uint32_t keep = a_ok ? 0xffffffffu : 0u;
      dst[0] = and4(ra0, keep); dst[1] = and4(ra1, keep); dst[2] = and4(ra2, keep); dst[3] = and4(ra3, keep);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
      *reinterpret_cast<uint4*>(&Bs[(b_row + i * 32) * LDS_PITCH + b_chunk * 8]) = and4(rb[i], b_ok[i] ? 0xffffffffu : 0u);
    __syncthreads();
    if (s + 1 < ksteps) SLAMHIP_IGEMM_ISSUE();   // in flight while the matrix cores work
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      bf16x8 af[2], bfr[NT];
#pragma unroll
      for (int i = 0; i < 2; ++i)
        af[i] = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 32 + (lane & 31)) * LDS_PITCH + kk + 8 * (lane >> 5)]);
#pragma unroll
      for (int j = 0; j < NT; ++j)
        bfr[j] = *reinterpret_cast<const bf16x8*>(&Bs[(wn * (BN / 2) + j * 32 + (lane & 31)) * LDS_PITCH + kk + 8 * (lane >> 5)]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
#undef SLAMHIP_IGEMM_ISSUE

  // C/D layout of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
  // A bfloat16-only output tile that lies wholly inside N leaves through LDS: rows of BN
  // elements written to memory as 16-byte chunks (BN / 8 lanes per row) instead of 64-byte runs
  // of 2-byte stores.
  const bool staged = WRITE_BF16 && !WRITE_F32 && n0 + BN <= N && (N & 7) == 0;
  if (staged) {
    constexpr int OP = BN + 8;                // pitch of the staged tile
    __syncthreads();   // the last step's fragments have been read
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          smem[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * OP + wn * (BN / 2) + j * 32 + (lane & 31)] =
              (bf16)acc[i][j][r];
    __syncthreads();
#pragma unroll
    for (int v = 0; v < BN / 16; ++v) {
      const int id = tid + 256 * v, row = id / (BN / 8), c8 = id % (BN / 8);
      if (m0 + row < M)
        *reinterpret_cast<uint4*>(out_bf16 + (long long)(m0 + row) * N + n0 + c8 * 8) =
            *reinterpret_cast<const uint4*>(&smem[row * OP + c8 * 8]);
    }
  }
#pragma unroll
  for (int i = 0; i < 2 && !staged; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int n = n0 + wn * (BN / 2) + j * 32 + (lane & 31);
        if (m < M && n < N) {
          const float v = acc[i][j][r];
          if (WRITE_BF16) out_bf16[(long long)m * N + n] = (bf16)v;
          if (WRITE_F32) {
            if (TAPS == 1 && splitk > 1) atomicAdd(&out_f32[(long long)m * N + n], v);
            else out_f32[(long long)m * N + n] = v;
          }
        }
      }
  // optional: this wave's column statistics (rows beyond M hold zeros and add nothing),
  // one record of 2N floats per (pixel tile, wave row)
  if (stat_part) {
    float* part = stat_part + ((long long)(work / n_tiles) * 2 + wm) * 2 * N;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = (float)(bf16)acc[i][j][r];
          sm += v;
          sq = __builtin_fmaf(v, v, sq);
        }
      store_column_stats(sm, sq, part, n0 + wn * (BN / 2) + j * 32 + (lane & 31), N, lane);
    }
  }
}

// ---------------------------------------------------------------------------
// 3x3 convolution (or data gradient) by filter row, for the layers with many pixels and few
// channels, where the per-tap kernel above spends its time staging: a K-step covers the three
// taps of one filter row and 64 channels.  The pixel tile is staged ONCE per step with one
// extra pixel on either side (pixels m0 - 1 .. m0 + 256 shifted by dy image rows) and the three
// taps read it at row offsets 0, 1, 2; whether a tap of a pixel lies inside the image is a
// per-lane bit (a fragment row IS a pixel), applied to the fragment in registers, so the staged
// rows need no masks, only addresses clamped into the tensor.  256 x 64 tiles: four waves
// stacked along the pixels, 64 x 64 outputs each (two A and two B fragments feed four MFMAs:
// 1 KB of LDS reads per MFMA, which bounds the 64-channel layers).  bfloat16 output; one record
// of column statistics per wave, i.e. two per 128 pixel rows as the other kernels leave.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256, 2)
igemm_row3w_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Bw, bf16* __restrict__ out_bf16, int M,
                        int N, int C, int H, int W, int flip, int n_tiles, int total_tiles,
                        float* __restrict__ stat_part, int m_tile0) {
  constexpr int BM = 256, BN = 64, AR = BM + 2;
  constexpr int NBV = 3 * BN / 32;            // 16-byte B loads per thread and step
#ifdef SLAMHIP_ABL_R3
  // timing-only ablations (wrong results): bit 0 no LDS stores, 1 no global loads after the first two steps,
  // 2 no output stores, 3 no fragment masks
  constexpr int ABL = SLAMHIP_ABL_R3;
#else
  constexpr int ABL = 0;
#endif
  __shared__ __attribute__((aligned(16))) bf16 As[(AR + 1) * LDS_PITCH];   // staged row AR holds zeros
  __shared__ __attribute__((aligned(16))) bf16 Bs[3 * BN * LDS_PITCH];

  const int per_xcd = (total_tiles + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_tiles) return;
  const int m0 = (m_tile0 + work / n_tiles) * BM, n0 = (work % n_tiles) * BN;   // m_tile0: first pixel tile of this launch

  const int tid = threadIdx.x, lane = tid & 63, wm = tid >> 6;
  // a tap outside the image reads the row of zeros instead of its pixel: one address select per fragment
  // row and tap in place of a mask on every fragment
  if (tid < LDS_PITCH / 8) *reinterpret_cast<uint4*>(&As[AR * LDS_PITCH + tid * 8]) = make_uint4(0u, 0u, 0u, 0u);
  const int Ktot = 9 * C;

  // A loader: eight passes; in pass v thread -> staged row (tid >> 3) + 32 v (pixel m0 - 1 + row),
  // 16-byte chunk tid & 7, so that eight lanes read one row's 128 bytes and a wave-instruction
  // touches 8 lines instead of 64; threads 0..15 also fetch one chunk of the two extra rows 256, 257
  const int a_row = tid >> 3, a_chunk = tid & 7;
  const int h_row = BM + (tid >> 3), h_chunk = tid & 7;
  const bool has_halo = tid < 16;
  // B loader: load v of a step brings weight slot e = v >> 1, output channel (tid >> 3) + 32 (v & 1), 16-byte
  // chunk tid & 7: one pointer and one LDS offset per thread, the rest are uniform offsets
  const bf16* const b_src0 = Bw + (long long)(n0 + (tid >> 3)) * Ktot + (tid & 7) * 8;
  const int b_dst0 = (tid >> 3) * LDS_PITCH + (tid & 7) * 8;
  const long long b_half = 32LL * Ktot;

  // validity of the nine taps of this lane's two fragment rows (pixels)
  unsigned int fm[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + wm * 64 + i * 32 + (lane & 31);
    unsigned int mask = 0u;
    if (m < M) {
      const int rem = m % (H * W);
      const int y = rem / W, x = rem - y * W;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((unsigned)(y + t / 3 - 1) < (unsigned)H && (unsigned)(x + t % 3 - 1) < (unsigned)W) mask |= 1u << t;
    }
    fm[i] = mask;
  }

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int chunks = C / 64, ksteps = 3 * chunks;
  // two register sets: the loads of two K-steps are in flight (a tile has as few as three steps,
  // and one step's MFMAs are shorter than a trip to HBM under load)
  uint4 ra[2][8], rah[2], rb[2][NBV];
  int g = 0, c0 = 0;     // filter row and channel chunk of the next loads to issue
  rah[0] = rah[1] = make_uint4(0u, 0u, 0u, 0u);
  const uint32_t b_keep = n0 < N ? 0xffffffffu : 0u;   // always set; the masked copy keeps the staged values in registers

#define SLAMHIP_ROW3W_ISSUE(set)                                                                 \
  {                                                                                              \
    const int dy = (flip ? 2 - g : g) - 1;                                                       \
    _Pragma("unroll") for (int v = 0; v < 8; ++v) {                                              \
      long long q = (long long)m0 - 1 + a_row + 32 * v + (long long)dy * W;                      \
      q = q < 0 ? 0 : (q > M - 1 ? M - 1 : q);                                                   \
      ra[set][v] = *reinterpret_cast<const uint4*>(A + q * C + c0 + a_chunk * 8);                \
    }                                                                                            \
    if (has_halo) {                                                                              \
      long long qh = (long long)m0 - 1 + h_row + (long long)dy * W;                              \
      qh = qh < 0 ? 0 : (qh > M - 1 ? M - 1 : qh);                                               \
      rah[set] = *reinterpret_cast<const uint4*>(A + qh * C + c0 + h_chunk * 8);                 \
    }                                                                                            \
    _Pragma("unroll") for (int v = 0; v < NBV; ++v)                                              \
      rb[set][v] = *reinterpret_cast<const uint4*>(b_src0 + (v & 1) * b_half + ((v >> 1) + g * 3) * C + c0); \
    c0 += 64;                                                                                    \
    if (c0 == C) { c0 = 0; ++g; }                                                                \
  }

  // fragments of sub-step it_ of the staged K-step into fragment set buf_ (staged row = pixel row + 1 + dx)
  bf16x8 fa[2][2], fb[2][2];
#define SLAMHIP_ROW3W_FRAGS(buf_, it_)                                                           \
  {                                                                                              \
    const int e_ = (it_) >> 2, kk_ = ((it_) & 3) * 16;                                           \
    const int shift_ = flip ? 2 - e_ : e_;                                                       \
    const int t_ = flip ? 8 - (g_cur * 3 + e_) : g_cur * 3 + e_;   /* geometric tap of weight slot (g_cur, e_) */ \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                              \
      const int row_ = (((fm[i] >> t_) & 1u) || (ABL & 8)) ? wm * 64 + i * 32 + (lane & 31) + shift_ : AR; \
      fa[buf_][i] = *reinterpret_cast<const bf16x8*>(&As[row_ * LDS_PITCH + kk_ + 8 * (lane >> 5)]); \
    }                                                                                            \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                \
      fb[buf_][j] = *reinterpret_cast<const bf16x8*>(                                            \
          &Bs[(e_ * BN + j * 32 + (lane & 31)) * LDS_PITCH + kk_ + 8 * (lane >> 5)]);            \
  }

  // one K-step from register set `set`; `more`: a step two ahead exists and is requested
#define SLAMHIP_ROW3W_STEP(set, more)                                                            \
  {                                                                                              \
    __syncthreads();   /* the previous step's fragments have been read */                        \
    if (!(ABL & 1) || s == 0) {                                                                  \
      _Pragma("unroll") for (int v = 0; v < 8; ++v)                                              \
        *reinterpret_cast<uint4*>(&As[(a_row + 32 * v) * LDS_PITCH + a_chunk * 8]) = and4(ra[set][v], b_keep); \
      if (has_halo) *reinterpret_cast<uint4*>(&As[h_row * LDS_PITCH + h_chunk * 8]) = rah[set];  \
      _Pragma("unroll") for (int v = 0; v < NBV; ++v)                                            \
        *reinterpret_cast<uint4*>(&Bs[b_dst0 + ((v >> 1) * BN + 32 * (v & 1)) * LDS_PITCH]) = and4(rb[set][v], b_keep); \
    }                                                                                            \
    __syncthreads();                                                                             \
    if ((more) && !(ABL & 2)) SLAMHIP_ROW3W_ISSUE(set);   /* in flight while the matrix cores work */ \
    /* twelve sub-steps (weight slot e, 16 channels): the fragments of sub-step it + 1 are read */ \
    /* while the MFMAs of sub-step it run (two fragment register sets) */                        \
    SLAMHIP_ROW3W_FRAGS(0, 0);                                                                   \
    _Pragma("unroll") for (int it = 0; it < 12; ++it) {                                          \
      if (it + 1 < 12) SLAMHIP_ROW3W_FRAGS((it + 1) & 1, it + 1);                                \
      __builtin_amdgcn_sched_barrier(0);   /* keep the reads ahead of this sub-step's MFMAs */    \
      _Pragma("unroll") for (int i = 0; i < 2; ++i)                                              \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                            \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it & 1][i], fb[it & 1][j], acc[i][j], 0, 0, 0); \
      __builtin_amdgcn_sched_barrier(0);                                                         \
    }                                                                                            \
    c_cur += 64;                                                                                 \
    if (c_cur == C) { c_cur = 0; ++g_cur; }                                                      \
  }

  int g_cur = 0, c_cur = 0;   // position of the step being computed
  SLAMHIP_ROW3W_ISSUE(0);
  if (ksteps > 1) SLAMHIP_ROW3W_ISSUE(1);
  for (int s = 0; s < ksteps; s += 2) {
    SLAMHIP_ROW3W_STEP(0, s + 2 < ksteps);
    if (s + 1 < ksteps) SLAMHIP_ROW3W_STEP(1, s + 3 < ksteps);
  }
#undef SLAMHIP_ROW3W_STEP
#undef SLAMHIP_ROW3W_FRAGS
#undef SLAMHIP_ROW3W_ISSUE

  if (ABL & 16) {   // no epilogue: one store that never happens keeps the accumulators alive
    float t = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[i][j][r];
    if (t == 12345.678f) out_bf16[tid] = (bf16)t;
    return;
  }
  // the 256 x 64 output tile leaves through LDS (the A tile's space): rows of 128 bytes, written
  // to memory as 16-byte chunks, eight lanes per row
  __syncthreads();   // the last step's fragments have been read
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        As[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDS_PITCH + j * 32 + (lane & 31)] =
            (bf16)acc[i][j][r];
  __syncthreads();
#pragma unroll
  for (int v = 0; v < 8; ++v) {
    const int id = tid + 256 * v, row = id >> 3, c8 = id & 7;
    if (m0 + row < M && (!(ABL & 4) || As[row * LDS_PITCH + c8 * 8] == (bf16)12345.f))
      *reinterpret_cast<uint4*>(out_bf16 + (long long)(m0 + row) * N + n0 + c8 * 8) =
          *reinterpret_cast<const uint4*>(&As[row * LDS_PITCH + c8 * 8]);
  }
  // records are counted in 64-row groups, two per 128 pixel rows; a wave wholly beyond the last
  // 128-row tile holds zeros and leaves no record
  if (!(ABL & 32) && stat_part && m0 / 64 + wm < (M + 127) / 128 * 2) {
    float* part = stat_part + ((long long)(work / n_tiles) * 4 + wm) * 2 * N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = (float)(bf16)acc[i][j][r];
          sm += v;
          sq = __builtin_fmaf(v, v, sq);
        }
      store_column_stats(sm, sq, part, n0 + j * 32 + (lane & 31), N, lane);
    }
  }
}

// ---------------------------------------------------------------------------
// The same implicit GEMM with 256 x 256 tiles for wide layers (N % 256 == 0): eight waves
// (2 x 4, 128 x 64 outputs each), both operand tiles of a K-step brought from global memory
// straight into LDS by global_load_lds_dwordx4 (no staging registers, no ds_write), two
// LDS buffers, one barrier per K-step: the loads of step s + 1 are in flight while the
// matrix cores work on step s.
//   LDS image of a tile: [256 rows][64 bf16] with 128-byte rows; a wave-instruction of the
//   DMA writes 1 KiB = 8 rows.  The 16-byte chunk q of row R is stored at position
//   q ^ ((R >> 1) & 7) (the loader permutes which global chunk a lane fetches), which makes
//   the fragment reads (ds_read_b128, 16 rows x one chunk per lane group) conflict-free.
//   Rows beyond M and taps outside the image fetch a line of zeros.
// ---------------------------------------------------------------------------
__device__ __attribute__((aligned(256))) uint4 g_zero_line[16];   // never written: zeros

typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;

// ---------------------------------------------------------------------------
// 3x3 convolution (or data gradient) of a 64 -> 64 channel layer with the WEIGHTS HELD IN REGISTERS: the whole
// 64 x 576 filter is 72 KB, i.e. 288 vector registers per lane as MFMA B fragments, which a wave keeps across a
// persistent walk over pixel tiles (one wave per SIMD, 512 registers each).  Only the pixel fragments come from
// LDS (0.5 KB of fragment reads per MFMA), the weights are read from memory once per workgroup, and the pixel
// tiles go global -> LDS through the DMA path (no staging registers, no ds_write).
//   A K-step is one filter row, as in the filter-row kernel above: pixels m0 - 1 .. m0 + 262 shifted by dy image
//   rows, 264 staged rows of 128 bytes (chunk q of row R at position q ^ ((R >> 1) & 7), permuted on the source
//   side), four LDS buffers so that the DMA loads of three steps are in flight; the three taps read the staged
//   tile at row offsets 0, 1, 2; a tap outside the image reads a row of zeros.  Four waves stacked along the
//   pixels, 64 x 64 outputs each; the output tile leaves through a wave-private LDS area in 16-byte stores.
//   Needs M * 128 < 2^32 (32-bit byte offsets) and H >= 256 / W + 1 (the pixel coordinates advance by a tile).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
conv3x3_c64_ws_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Bw, bf16* __restrict__ out_bf16, int M,
                      int H, int W, int flip, int total_tiles, float* __restrict__ stat_part) {
  constexpr int C = 64, N = 64, ROWS = 264, BUF = ROWS * 128, OUT_AREA = 32 * 128, NBUF = 4;
#ifdef SLAMHIP_ABL_WS     // timing-only ablations (wrong results): 1 no epilogue, 2 no DMA loads after the first three, 4 no fragment reads
  constexpr int ABL = SLAMHIP_ABL_WS;
#else
  constexpr int ABL = 0;
#endif
  __shared__ __attribute__((aligned(1024))) unsigned char lds[NBUF * BUF + 1024];
  // (an array of its own: accesses to it are known not to touch what the DMA loads in flight write)
  __shared__ __attribute__((aligned(1024))) unsigned char lds_out[4 * OUT_AREA];
  constexpr int ZERO_OFF = NBUF * BUF;   // a row of zeros

  // contiguous runs of tiles per workgroup, neighbouring runs on one XCD (the halo rows of a tile are its
  // neighbours' rows)
  const int nwg = gridDim.x, per = (total_tiles + nwg - 1) / nwg;
  const int wg = (blockIdx.x & 7) * (nwg >> 3) + (blockIdx.x >> 3);
  const int t_begin = wg * per, t_end = min(t_begin + per, total_tiles);
  if (t_begin >= t_end) return;

  const int tid = threadIdx.x, lane = tid & 63, wm = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qh = lane >> 5;
  const unsigned int lds_base = (unsigned int)(uintptr_t)(lds_void_ptr)lds;
  if (tid < 8) *reinterpret_cast<uint4*>(lds + ZERO_OFF + tid * 16) = make_uint4(0u, 0u, 0u, 0u);

  // the filter as B fragments: weight slot e (0..8), 16-channel step kk, output half j
  bf16x8 fb[9][4][2];
#pragma unroll
  for (int e = 0; e < 9; ++e)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int j = 0; j < 2; ++j)
        fb[e][kk][j] = *reinterpret_cast<const bf16x8*>(Bw + (long long)(j * 32 + (lane & 31)) * (9 * C) + e * C + kk * 16 + 8 * qh);

  // requests the staged rows of filter row g of tile `tile` into buffer `buf`: piece P (8 rows, 1 KiB) = wave * 8 + j,
  // the 33rd piece by wave 0; lane -> row 8 P + (lane >> 3), position lane & 7
  auto issue = [&](const int tile, const int g, const int buf) {
    int ln = lane;
    asm volatile("" : "+v"(ln));     // (a lane's row and chunk are made again at every call: held across the tile
                                     // loop they take registers the filter needs)
    const int dy = (flip ? 2 - g : g) - 1;
    const int rl = ln >> 3;
    const int q0 = tile * 256 - 1 + dy * W + wm * 64 + rl;             // pixel of piece wave * 8, this lane's row
    const unsigned int c_even = (unsigned int)(((ln & 7) ^ (rl >> 1)) << 4), c_odd = c_even ^ 64u;   // (R >> 1) & 7 by piece parity
    unsigned char* dst = lds + buf * BUF + wm * 8192;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int q = min(max(q0 + 8 * j, 0), M - 1);
      const char* src = reinterpret_cast<const char*>(A) + (size_t)(((unsigned int)q << 7) + ((j & 1) ? c_odd : c_even));
      __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)(dst + j * 1024), 16, 0, 0);
    }
    if (wm == 0) {
      const int q = min(max(q0 + 256, 0), M - 1);
      const char* src = reinterpret_cast<const char*>(A) + (size_t)(((unsigned int)q << 7) + c_even);
      __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)(dst + 32 * 1024), 16, 0, 0);
    }
  };
  // waits until this wave's loads of all but the newest `newer` steps have landed (stores in flight only make
  // the wait stricter: the counter counts them too)
  auto wait_loads = [&](const int newer) {
    if (newer <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if (newer == 1) { if (wm == 0) asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); }
    else { if (wm == 0) asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); }
  };

  const int n_steps = (t_end - t_begin) * 3;
  __syncthreads();     // the row of zeros
  issue(t_begin, 0, 0);
  issue(t_begin, 1, 1);
  issue(t_begin, 2, 2);

  // image coordinates of this lane's two fragment rows (pixels), advanced by 256 pixels per tile
  const int step_x = 256 % W, step_y = 256 / W;
  int px[2], py[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const long long m = (long long)t_begin * 256 + wm * 64 + i * 32 + (lane & 31);
    const int rem = (int)(m % ((long long)H * W));
    py[i] = rem / W;
    px[i] = rem - py[i] * W;
  }

  int step = 0;
#pragma unroll 1
  for (int tile = t_begin; tile < t_end; ++tile) {
    const int m0 = tile * 256;
    int ln = lane;
    asm volatile("" : "+v"(ln));     // (opaque per tile: nothing below is hoisted out of the tile loop into registers)
    const int l31 = ln & 31, qo = ln >> 5;
    // validity of the nine taps of the two pixels
    unsigned int fm[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = m0 + wm * 64 + i * 32 + l31, y = py[i], x = px[i];
      unsigned int mask = 0u;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((unsigned)(y + t / 3 - 1) < (unsigned)H && (unsigned)(x + t % 3 - 1) < (unsigned)W) mask |= 1u << t;
      fm[i] = m < M ? mask : 0u;
      int nx = x + step_x, ny = y + step_y;
      if (nx >= W) { nx -= W; ++ny; }
      if (ny >= H) ny -= H;
      px[i] = nx; py[i] = ny;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

#pragma unroll
    for (int g = 0; g < 3; ++g, ++step) {
      // this step's rows have landed (every wave's: the barrier), and every wave has finished with the buffer
      // that the loads issued below overwrite (it held the step before this one)
      if (!(ABL & 2) || step < 3) wait_loads(min(n_steps - 1 - step, 2));
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();     // (a raw barrier: __syncthreads would also wait for the loads in flight)
      __builtin_amdgcn_sched_barrier(0);
      if (step + 3 < n_steps && !(ABL & 2)) issue(tile + 1, g, (step + 3) & 3);
      const unsigned int buf_base = lds_base + (unsigned int)((step & 3) * BUF);
      // The fragment reads are written as assembly with their own waits: the compiler takes every LDS read behind a
      // DMA load for a possible reader of what that load writes and would wait for all loads in flight (the
      // barrier above is what orders them).
      bf16x8 fa[2][2];
      // per weight slot e of this filter row and fragment row i: the staged row's address (or the row of zeros when
      // the tap lies outside the image) and the swizzled position of 16-channel step 0; step kk is position ^ (kk << 5)
      unsigned int rb[3][2], cp[3][2];
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int shift = flip ? 2 - e : e;
        const int t = flip ? 8 - (g * 3 + e) : g * 3 + e;   // geometric tap of weight slot (g, e)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int R = wm * 64 + i * 32 + l31 + shift;
          const unsigned int row_addr = buf_base + (unsigned int)(R * 128), zero_addr = lds_base + (unsigned int)ZERO_OFF;
          rb[e][i] = ((fm[i] >> t) & 1u) ? row_addr : zero_addr;
          cp[e][i] = (unsigned int)((qo ^ ((R >> 1) & 7)) << 4);
        }
      }
      auto frags = [&](const int set, const int it) {
        const int e = it >> 2, kk = it & 3;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const unsigned int addr = (cp[e][i] ^ (unsigned int)(kk << 5)) + rb[e][i];
          if (ABL & 4) asm volatile("; no fragment read %0 %1" : "=v"(fa[set][i]) : "v"(addr));
          else asm volatile("ds_read_b128 %0, %1" : "=v"(fa[set][i]) : "v"(addr));
        }
      };
      frags(0, 0);
#pragma unroll
      for (int it = 0; it < 12; ++it) {
        // LDS reads of one wave return in order: "at most two outstanding" means that the pair read one sub-step
        // earlier has arrived
        if (it + 1 < 12) {
          frags((it + 1) & 1, it + 1);
          asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa[it & 1][0]), "+v"(fa[it & 1][1]));
        } else {
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[it & 1][0]), "+v"(fa[it & 1][1]));
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[it & 1][i], fb[g * 3 + (it >> 2)][it & 3][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    if (ABL & 1) {   // one store that never happens keeps the accumulators alive
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) t += acc[i][j][r];
      if (t == 12345.678f) out_bf16[tid] = (bf16)t;
      continue;
    }
    // the wave's 64 x 64 outputs through its own LDS area, 32 rows of 128 bytes at a time, then 16-byte stores
    unsigned char* oa = lds_out + wm * OUT_AREA;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * qo, colc = j * 32 + l31;
          // (chunk swizzle by row so that the 16-byte reads of a row and these 2-byte writes spread over the banks)
          *reinterpret_cast<bf16*>(oa + row * 128 + ((((colc >> 3) ^ (row & 7)) << 4) | ((colc & 7) << 1))) = (bf16)acc[i][j][r];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int row = v * 8 + (ln >> 3), c8 = ln & 7;
        const int m = m0 + wm * 64 + i * 32 + row;
        const uint4 val = *reinterpret_cast<const uint4*>(oa + row * 128 + ((c8 ^ (row & 7)) << 4));
        if (m < M) *reinterpret_cast<uint4*>(out_bf16 + (long long)m * N + c8 * 8) = val;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    if (stat_part && m0 / 64 + wm < (M + 127) / 128 * 2) {
      float* part = stat_part + ((long long)tile * 4 + wm) * 2 * N;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float sm = 0.f, sq = 0.f;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float v = (float)(bf16)acc[i][j][r];
            sm += v;
            sq = __builtin_fmaf(v, v, sq);
          }
        store_column_stats(sm, sq, part, j * 32 + l31, N, ln);
      }
    }
  }
}

template <int TAPS>
__global__ void __launch_bounds__(512)
igemm256_bf16_kernel(const bf16* __restrict__ A, const bf16* __restrict__ Bw, bf16* __restrict__ out_bf16, int M,
                     int N, int C, int H, int W, int flip, int n_tiles, int total_tiles,
                     float* __restrict__ stat_part, int m_tile0) {
  constexpr int TILE_BYTES = 256 * 128;                         // one operand tile of one K-step
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * TILE_BYTES];   // [buffer][A | B]

  const int per_xcd = (total_tiles + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_tiles) return;
  const int m0 = (m_tile0 + work / n_tiles) * 256, n0 = (work % n_tiles) * 256;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int Ktot = TAPS * C;

  // loader: piece P = wave * 4 + j (j = 0..3) covers rows 8P .. 8P + 7; lane -> row 8P + (lane >> 3),
  // position lane & 7, fetching global chunk position ^ ((row >> 1) & 7)
  const bf16* a_src[4];
  const bf16* b_src[4];
  unsigned int a_taps[4];   // bit t: the row exists and tap t of it lies inside the image
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 8 + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    const bool row_ok = m0 + r < M;
    const int pm = row_ok ? m0 + r : M - 1;
    a_taps[j] = row_ok ? 1u : 0u;
    if (TAPS == 9) {
      const int rem = pm % (H * W);
      const int y = rem / W, x = rem - y * W;
      unsigned int mask = 0u;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((unsigned)(y + t / 3 - 1) < (unsigned)H && (unsigned)(x + t % 3 - 1) < (unsigned)W) mask |= 1u << t;
      a_taps[j] = row_ok ? mask : 0u;
    }
    a_src[j] = A + (long long)pm * C + g * 8;
    b_src[j] = Bw + (long long)(n0 + r) * Ktot + g * 8;
  }
  const bf16* zero_src = reinterpret_cast<const bf16*>(g_zero_line);

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  const int ksteps = TAPS * (C / 64);
  int tap = 0, c0 = 0;   // position of the next tile to request

  // request the tile (tap, c0) into buffer `buf` and advance the position
#define SLAMHIP_IGEMM256_ISSUE(buf)                                                                  \
  {                                                                                                  \
    int t = 0;                                                                                       \
    long long delta = c0;                                                                            \
    if (TAPS == 9) {                                                                                 \
      t = flip ? 8 - tap : tap;                                                                      \
      delta += (long long)((t / 3 - 1) * W + (t - (t / 3) * 3 - 1)) * C;                             \
    }                                                                                                \
    unsigned char* a_dst = lds + (buf) * 2 * TILE_BYTES + wave * 4096;                               \
    unsigned char* b_dst = a_dst + TILE_BYTES;                                                       \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                  \
      const bf16* src = ((a_taps[j] >> t) & 1u) ? a_src[j] + delta : zero_src;                       \
      __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)(a_dst + j * 1024), 16, 0, 0); \
    }                                                                                                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                    \
      __builtin_amdgcn_global_load_lds((global_void_ptr)(b_src[j] + tap * C + c0),                   \
                                       (lds_void_ptr)(b_dst + j * 1024), 16, 0, 0);                  \
    c0 += 64;                                                                                        \
    if (c0 == C) { c0 = 0; ++tap; }                                                                  \
  }

  // fragment addresses (bytes within a tile): row R, chunk q -> R * 128 + ((q ^ ((R >> 1) & 7)) << 4)
  int a_off[4], b_off[2];
  const int qh = lane >> 5;   // which 8-element half of a 16-deep MFMA step
#pragma unroll
  for (int i = 0; i < 4; ++i) { const int R = wm * 128 + i * 32 + (lane & 31); a_off[i] = R * 128; }
#pragma unroll
  for (int j = 0; j < 2; ++j) { const int R = wn * 64 + j * 32 + (lane & 31); b_off[j] = R * 128; }
  const int swz = (((lane & 31) >> 1) & 7);   // (R >> 1) & 7: the tile bases are multiples of 32 rows

  // two of the eight DMA loads of a tile: parts 0, 1 = operand A pieces 0-1, 2-3; parts 2, 3 = operand B
#define SLAMHIP_IGEMM256_PART(buf, part)                                                             \
  {                                                                                                  \
    unsigned char* a_dst = lds + (buf) * 2 * TILE_BYTES + wave * 4096;                               \
    unsigned char* b_dst = a_dst + TILE_BYTES;                                                       \
    if ((part) < 2) {                                                                                \
      int t = 0;                                                                                     \
      long long delta = c0;                                                                          \
      if (TAPS == 9) {                                                                               \
        t = flip ? 8 - tap : tap;                                                                    \
        delta += (long long)((t / 3 - 1) * W + (t - (t / 3) * 3 - 1)) * C;                           \
      }                                                                                              \
      _Pragma("unroll") for (int j = 2 * (part); j < 2 * (part) + 2; ++j) {                          \
        const bf16* src = ((a_taps[j] >> t) & 1u) ? a_src[j] + delta : zero_src;                     \
        __builtin_amdgcn_global_load_lds((global_void_ptr)src, (lds_void_ptr)(a_dst + j * 1024), 16, 0, 0); \
      }                                                                                              \
    } else {                                                                                         \
      _Pragma("unroll") for (int j = 2 * (part) - 4; j < 2 * (part) - 2; ++j)                        \
        __builtin_amdgcn_global_load_lds((global_void_ptr)(b_src[j] + tap * C + c0),                 \
                                         (lds_void_ptr)(b_dst + j * 1024), 16, 0, 0);                \
    }                                                                                                \
  }
#define SLAMHIP_IGEMM256_ADVANCE() \
  {                                \
    c0 += 64;                      \
    if (c0 == C) { c0 = 0; ++tap; } \
  }

  {
    // The two wave rows (wm = 0 / 1, one wave of each per SIMD) run half a phase apart: while one
    // row's sixteen MFMAs of a phase occupy the matrix cores, the other row reads its next
    // fragments from LDS and, in the first phase of a tile, issues its share of the next tile's
    // DMA loads.  Every section ends in a raw s_barrier, so the two rows alternate strictly; each
    // wave executes the same number of barriers (the extra one in front for row 1, behind for
    // row 0).  Two phases per tile: rows 0-63 of the wave's 128 x 64 outputs with all 64 columns
    // (8 A and 8 B fragments), then rows 64-127 with the same B fragments (8 A fragments).
    // A load section waits for its own LDS reads (and, in the second phase of a tile, for its DMA
    // loads of the next tile) BEFORE its barrier: a buffer is re-filled only after every read of
    // it has returned, and read only after every wave's loads into it have landed.
#define SLAMHIP_PP_SECTION_END()                          \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                      \
  __builtin_amdgcn_s_barrier();                           \
  __builtin_amdgcn_sched_barrier(0);
#define SLAMHIP_PP_LOAD_A(ih)                                                                         \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk) {                                                  \
    const int chunk = ((kk * 2 + qh) ^ swz) << 4;                                                     \
    _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                     \
      af[i][kk] = *reinterpret_cast<const bf16x8*>(a_tile + a_off[(ih) * 2 + i] + chunk);             \
  }
#define SLAMHIP_PP_LOAD_B(dst, j)                                                                     \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                    \
    dst[kk] = *reinterpret_cast<const bf16x8*>(b_tile + b_off[j] + ((((kk * 2 + qh) ^ swz)) << 4));
#define SLAMHIP_PP_MFMA2(ih)                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                      \
  _Pragma("unroll") for (int kk = 0; kk < 4; ++kk)                                                    \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                   \
      acc[(ih) * 2 + i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][kk], bq0[kk], acc[(ih) * 2 + i][0], 0, 0, 0); \
      acc[(ih) * 2 + i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i][kk], bq1[kk], acc[(ih) * 2 + i][1], 0, 0, 0); \
    }                                                                                                 \
  __builtin_amdgcn_s_setprio(0);                                                                      \
  __builtin_amdgcn_sched_barrier(0);                                                                  \
  __builtin_amdgcn_s_barrier();                                                                       \
  __builtin_amdgcn_sched_barrier(0);

    SLAMHIP_IGEMM256_PART(0, 0);
    SLAMHIP_IGEMM256_PART(0, 1);
    SLAMHIP_IGEMM256_PART(0, 2);
    SLAMHIP_IGEMM256_PART(0, 3);
    SLAMHIP_IGEMM256_ADVANCE();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wm == 1) __builtin_amdgcn_s_barrier();
    bf16x8 af[2][4], bq0[4], bq1[4];
    for (int s = 0; s < ksteps; ++s) {
      const unsigned char* a_tile = lds + (s & 1) * 2 * TILE_BYTES;
      const unsigned char* b_tile = a_tile + TILE_BYTES;
      const bool more = s + 1 < ksteps;
      const int nbuf = (s + 1) & 1;
      // phase 1: rows 0-63 of the wave x all 64 columns (16 MFMAs); the whole next tile is requested here
      SLAMHIP_PP_LOAD_A(0);
      SLAMHIP_PP_LOAD_B(bq0, 0);
      SLAMHIP_PP_LOAD_B(bq1, 1);
      if (more) {
        SLAMHIP_IGEMM256_PART(nbuf, 0); SLAMHIP_IGEMM256_PART(nbuf, 1);
        SLAMHIP_IGEMM256_PART(nbuf, 2); SLAMHIP_IGEMM256_PART(nbuf, 3);
        SLAMHIP_IGEMM256_ADVANCE();
      }
      SLAMHIP_PP_SECTION_END();
      SLAMHIP_PP_MFMA2(0);
      // phase 2: rows 64-127 with the same B fragments
      SLAMHIP_PP_LOAD_A(1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      SLAMHIP_PP_SECTION_END();
      SLAMHIP_PP_MFMA2(1);
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
#undef SLAMHIP_PP_SECTION_END
#undef SLAMHIP_PP_LOAD_A
#undef SLAMHIP_PP_LOAD_B
#undef SLAMHIP_PP_MFMA2
  }
#undef SLAMHIP_IGEMM256_ISSUE
#undef SLAMHIP_IGEMM256_PART
#undef SLAMHIP_IGEMM256_ADVANCE

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int n = n0 + wn * 64 + j * 32 + (lane & 31);
        if (m < M) out_bf16[(long long)m * N + n] = (bf16)acc[i][j][r];
      }
  if (stat_part) {   // one record of 2N floats per (256-row tile, wave row)
    float* part = stat_part + ((long long)(work / n_tiles) * 2 + wm) * 2 * N;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = (float)(bf16)acc[i][j][r];
          sm += v;
          sq = __builtin_fmaf(v, v, sq);
        }
      store_column_stats(sm, sq, part, n0 + wn * 64 + j * 32 + (lane & 31), N, lane);
    }
  }
}

// ---------------------------------------------------------------------------
// weight gradient:  dW[tap][ci][co] += sum_p X[pixel(p) + tap][ci] * dY[p][co]
// (TAPS = 1: dW[ci][co] = X^T dY over the rows).  Both operands sit pixel-major
// in LDS exactly as they come from HBM; the MFMA fragments (8 consecutive pixels
// of one channel) are read with the transposing ds_read_b64_tr_b16.
// A workgroup owns a TM x TN (ci x co) tile of one tap and one slice of the
// pixels; four waves split it 2 x 2.  Slices are dealt to the XCDs in contiguous
// runs so that all tiles and taps of a slice read it through one L2.  With more
// than one slice the partial tiles are added with float32 atomics into the
// zeroed dW.
// ---------------------------------------------------------------------------
template <int PITCH>
__device__ __forceinline__ bf16x8 tr_frag(const bf16* tile, int k0, int mbase, int lane) {
  // lanes 16g..16g+15: block of 4 pixel rows x 16 channels; lane 4q+p gives row q, channels 4p..4p+3
  const int g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3, h = g >> 1;
  const bf16* a0 = tile + (k0 + 8 * h + q) * PITCH + mbase + (g & 1) * 16 + 4 * p;
  const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)a0);
  const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(a0 + 4 * PITCH));
  bf16x8 f;
  const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
  for (int j = 0; j < 4; ++j) { f[j] = l4[j]; f[4 + j] = h4[j]; }
  return f;
}

template <int TAPS, int TM, int TN>
__global__ void __launch_bounds__(256)
wgrad_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ dW, int P, int Cin,
                  int Cout, int H, int W, int splitk, int total_work, long long det_stride) {
  // row pitch = payload + 64 bytes: the 4 rows x 2 column halves that 32 lanes of a
  // transposing read touch then fall into 8 distinct 32-byte bank groups
  constexpr int PX = TM + 32, PY = TN + 32;
  constexpr int MT = TM / 64, NT = TN / 64;     // 32x32 MFMA tiles per wave along ci / co
  constexpr int XV = TM / 32, YV = TN / 32;     // 16-byte loads per thread per 64-pixel step
  __shared__ __attribute__((aligned(16))) bf16 Xs[64 * PX];
  __shared__ __attribute__((aligned(16))) bf16 Ys[64 * PY];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int per_xcd = (total_work + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_work) return;
  const int ci_tiles = Cin / TM, co_tiles = Cout / TN, tiles = ci_tiles * co_tiles * TAPS;
  const int slice = work / tiles;
  int rest = work - slice * tiles;
  const int tap = rest / (ci_tiles * co_tiles);
  rest -= tap * ci_tiles * co_tiles;
  const int ci0 = (rest / co_tiles) * TM, co0 = (rest % co_tiles) * TN;
  const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap % 3 - 1 : 0;
  // slices are multiples of 64 pixels so that every step but the last is full
  const int chunk = (((P + splitk - 1) / splitk) + 63) & ~63;
  const int p_begin = slice * chunk, p_end = min(P, p_begin + chunk);

  // loader: thread -> pixel row tid>>2 of the 64-pixel step, quarter tid&3 of the channels.
  // Rows past the slice and taps outside the image read a valid dummy address and are
  // zeroed by mask.
  const int l_row = tid >> 2, l_q = tid & 3;
  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  uint4 rx[XV], ry[YV];
  uint32_t keep_x = 0u, keep_y = 0u;
  const int hw = H * W;
  const bf16* x_col = X + ci0 + l_q * (TM / 4);
  const bf16* y_col = dY + co0 + l_q * (TN / 4);

  // Pixel coordinates advance with the steps: (y0, x0) belongs to the step's first pixel and a
  // lane adds its row; floor(x / W) for x < W + 64 is a compare (W >= 64) or one multiply by
  // a 20-bit reciprocal (exact while x * W < 2^20, here x < 128).
  const unsigned int magic_w = ((1u << 20) + (unsigned int)W - 1u) / (unsigned int)W;
  int y0 = 0, x0 = 0;
  if (TAPS == 9) {
    const int rem0 = p_begin % hw;
    y0 = rem0 / W;
    x0 = rem0 - y0 * W;
  }
#define SLAMHIP_WGRAD_ISSUE(p0)                                                                     \
  {                                                                                                 \
    const int p = (p0) + l_row;                                                                     \
    const bool in_slice = p < p_end;                                                                \
    const int pc = in_slice ? p : p_end - 1;                                                        \
    long long xo = (long long)pc * Cin;                                                             \
    bool x_ok = in_slice;                                                                           \
    if (TAPS == 9) {                                                                                \
      int x = x0 + l_row;                                                                           \
      const int q = (W >= 64) ? (x >= W ? 1 : 0) : (int)(((unsigned int)x * magic_w) >> 20);        \
      x -= q * W;                                                                                   \
      int y = y0 + q;                                                                               \
      while (y >= H) y -= H;                                                                        \
      x_ok = in_slice && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W;      \
      if (x_ok) xo += (long long)(dy * W + dx) * Cin;                                               \
    }                                                                                               \
    const uint4* xs = reinterpret_cast<const uint4*>(x_col + xo);                                   \
    const uint4* ys = reinterpret_cast<const uint4*>(y_col + (long long)pc * Cout);                 \
    _Pragma("unroll") for (int i = 0; i < XV; ++i) rx[i] = xs[i];                                   \
    _Pragma("unroll") for (int i = 0; i < YV; ++i) ry[i] = ys[i];                                   \
    keep_x = x_ok ? 0xffffffffu : 0u;                                                               \
    keep_y = in_slice ? 0xffffffffu : 0u;                                                           \
    if (TAPS == 9) {                                                                                \
      x0 += 64;                                                                                     \
      const int q0 = (W >= 64) ? (x0 >= W ? 1 : 0) : (int)(((unsigned int)x0 * magic_w) >> 20);     \
      x0 -= q0 * W;                                                                                 \
      y0 += q0;                                                                                     \
      while (y0 >= H) y0 -= H;                                                                      \
    }                                                                                               \
  }

  if (p_begin < p_end) SLAMHIP_WGRAD_ISSUE(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    __syncthreads();
    {
      uint4* xd = reinterpret_cast<uint4*>(&Xs[l_row * PX + l_q * (TM / 4)]);
      uint4* yd = reinterpret_cast<uint4*>(&Ys[l_row * PY + l_q * (TN / 4)]);
#pragma unroll
      for (int i = 0; i < XV; ++i) xd[i] = and4(rx[i], keep_x);
#pragma unroll
      for (int i = 0; i < YV; ++i) yd[i] = and4(ry[i], keep_y);
    }
    __syncthreads();
    if (p0 + 64 < p_end) SLAMHIP_WGRAD_ISSUE(p0 + 64);
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      bf16x8 af[MT], bfr[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) af[i] = tr_frag<PX>(Xs, kk, wm * (TM / 2) + i * 32, lane);    // A[m = ci][k = pixel]
#pragma unroll
      for (int j = 0; j < NT; ++j) bfr[j] = tr_frag<PY>(Ys, kk, wn * (TN / 2) + j * 32, lane);   // B[k = pixel][n = co]
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  }
#undef SLAMHIP_WGRAD_ISSUE
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = co0 + wn * (TN / 2) + j * 32 + (lane & 31);
        float* dst = &dW[((long long)tap * Cin + ci) * Cout + co];
        if (splitk == 1) *dst = acc[i][j][r];   // the only contribution: plain store, dW needs no zero-fill
        else if (det_stride) dst[slice * det_stride] = acc[i][j][r];   // fixed-order mode: this slice's own slot
        else atomicAdd(dst, acc[i][j][r]);
      }
}

// ---------------------------------------------------------------------------
// Filter gradient with 256 x 256 (ci x co) tiles for the wide layers (Cin, Cout multiples of 256): the operand
// scheme of igemm256_bf16_kernel - both tiles of a K-step go from global memory straight into LDS by
// global_load_lds_dwordx4, two LDS buffers, eight waves (2 x 4, 128 x 64 outputs each), 0.75 KB of fragment reads
// per MFMA - with the pixels as the K dimension: a K-step is 64 consecutive pixels, X and dY stay pixel-major in LDS
// ([64 pixels][256 channels], rows of 512 bytes) and the fragments are read transposed (ds_read_b64_tr_b16).
// A workgroup owns one tap of one tile and one slice of the pixels.
//   LDS image of a tile: a DMA wave-instruction writes 1 KiB = two pixel rows; the 16-byte chunk q of row R is
//   stored at position q ^ ((R & 3) << 2) (the loader permutes which global chunk a lane fetches), so that the four
//   pixel rows x 64 bytes that 32 lanes of a transposing read touch fall into four distinct 64-byte bank groups.
//   Pixels past the slice and taps outside the image fetch a line of zeros.
// Pixel coordinates: row index q = p / W and q mod H by multiplication with 32-bit reciprocals (exact while
// P * W < 2^32, checked by the caller).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
wgrad256_bf16_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ dW, int P, int Cin,
                     int Cout, int H, int W, int splitk, int total_work, unsigned int magic_w, unsigned int magic_h,
                     long long det_stride) {
  constexpr int TILE_BYTES = 64 * 512;                          // one operand tile of one K-step
  __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * TILE_BYTES];   // [buffer][X | dY]

  const int per_xcd = (total_work + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_work) return;
  const int ci_tiles = Cin >> 8, co_tiles = Cout >> 8, jobs = 9 * ci_tiles * co_tiles;
  const int slice = work / jobs;
  int rest = work - slice * jobs;
  const int tap = rest / (ci_tiles * co_tiles);
  rest -= tap * ci_tiles * co_tiles;
  const int ci0 = (rest / co_tiles) << 8, co0 = (rest % co_tiles) << 8;
  const int dy = tap / 3 - 1, dx = tap % 3 - 1;
  const int chunk = (((P + splitk - 1) / splitk) + 63) & ~63;   // slices are multiples of 64 pixels
  const int p_begin = slice * chunk, p_end = min(P, p_begin + chunk);
  const int ksteps = (p_end - p_begin + 63) >> 6;               // >= 1 (the caller drops empty slices)

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;

  // loader: piece j (0..3) of this wave covers the pixel rows r = (wave * 4 + j) * 2 + (lane >> 5) of the K-step;
  // a lane sits at position lane & 31 and fetches global chunk position ^ ((r & 3) << 2)
  int l_row[4], l_chunk[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    l_row[j] = (wave * 4 + j) * 2 + (lane >> 5);
    l_chunk[j] = (lane & 31) ^ ((l_row[j] & 3) << 2);
  }
  const bf16* const x_base = X + ci0;
  const bf16* const y_base = dY + co0;
  const bf16* const zero_src = reinterpret_cast<const bf16*>(g_zero_line);
  const long long tap_off = (long long)(dy * W + dx) * Cin;

  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // request the tiles of K-step s into buffer buf
#define SLAMHIP_WG256_ISSUE(buf, s_)                                                                  \
  {                                                                                                   \
    unsigned char* x_dst = lds + (buf) * 2 * TILE_BYTES + wave * 4096;                                \
    unsigned char* y_dst = x_dst + TILE_BYTES;                                                        \
    const int pk = p_begin + ((s_) << 6);                                                             \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) {                                                   \
      const int pp = pk + l_row[j];                                                                   \
      const bool in_slice = pp < p_end;                                                               \
      const unsigned int q = __umulhi((unsigned int)pp, magic_w);      /* image row over the batch */ \
      const int x = pp - (int)q * W;                                                                  \
      const int y = (int)q - (int)__umulhi(q, magic_h) * H;                                           \
      const bool ok = in_slice && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W; \
      const bf16* xs = ok ? x_base + (long long)pp * Cin + tap_off + l_chunk[j] * 8 : zero_src;       \
      const bf16* ys = in_slice ? y_base + (long long)pp * Cout + l_chunk[j] * 8 : zero_src;          \
      __builtin_amdgcn_global_load_lds((global_void_ptr)xs, (lds_void_ptr)(x_dst + j * 1024), 16, 0, 0); \
      __builtin_amdgcn_global_load_lds((global_void_ptr)ys, (lds_void_ptr)(y_dst + j * 1024), 16, 0, 0); \
    }                                                                                                 \
  }

  // transposing fragment read from a swizzled tile: 16 pixels k0 .. k0 + 15 x 32 channels mbase ...
  auto frag = [&](const unsigned char* tile, int k0, int mbase) -> bf16x8 {
    // lanes 16g..16g+15: block of 4 pixel rows x 16 channels; lane 4q+p gives row q, channels 4p..4p+3
    const int g = lane >> 4, i = lane & 15, q = i >> 2, p4 = i & 3, h = g >> 1;
    const int c = mbase + (g & 1) * 16 + 4 * p4;                 // first of the lane's four channels
    const int R0 = k0 + 8 * h + q, R1 = R0 + 4;                  // (R0 & 3) == (R1 & 3) == q: one swizzle for both
    const int off = (((c >> 3) ^ (q << 2)) << 4) + ((c & 7) << 1);
    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(tile + R0 * 512 + off));
    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((short4_t __attribute__((address_space(3)))*)(tile + R1 * 512 + off));
    bf16x8 f;
    const bf16x4 l4 = __builtin_bit_cast(bf16x4, lo), h4 = __builtin_bit_cast(bf16x4, hi);
#pragma unroll
    for (int j = 0; j < 4; ++j) { f[j] = l4[j]; f[4 + j] = h4[j]; }
    return f;
  };

  // One barrier per K-step: the next step's tiles are requested, this step's fragments read and multiplied, then the
  // wave waits for its own DMA loads and the barrier hands the buffers over.  (The two-phase schedule of
  // igemm256_bf16_kernel, wave rows half a phase apart, was measured here too: 4.5 % faster at 56 x 56 x 256 x 256,
  // 2-3 % slower on the three 28 x 28 and 14 x 14 shapes.)
  SLAMHIP_WG256_ISSUE(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int s = 0; s < ksteps; ++s) {
    const unsigned char* x_tile = lds + (s & 1) * 2 * TILE_BYTES;
    const unsigned char* y_tile = x_tile + TILE_BYTES;
    if (s + 1 < ksteps) SLAMHIP_WG256_ISSUE((s + 1) & 1, s + 1);   // in flight while the matrix cores work
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      bf16x8 af[4], bfr[2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[i] = frag(x_tile, kk, wm * 128 + i * 32);      // A[m = ci][k = pixel]
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[j] = frag(y_tile, kk, wn * 64 + j * 32);      // B[k = pixel][n = co]
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    // the next step's tiles have landed (this wave's DMA) and every wave is done reading this step's buffer
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#undef SLAMHIP_WG256_ISSUE

#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = co0 + wn * 64 + j * 32 + (lane & 31);
        float* dst = &dW[((long long)tap * Cin + ci) * Cout + co];
        if (splitk == 1) *dst = acc[i][j][r];
        else if (det_stride) dst[slice * det_stride] = acc[i][j][r];   // fixed-order mode: this slice's own slot
        else atomicAdd(dst, acc[i][j][r]);
      }
}

// ---------------------------------------------------------------------------
// Filter gradient of the layers with wide images and few channels: the per-tap
// kernel above reads X and dY nine times from L2 and is bound by that.  Here a workgroup owns
// one filter ROW (dy fixed, dx = -1, 0, +1) of a 64 x 64 (ci x co) tile: a K-step is a segment
// of 112 consecutive pixels of ONE image row, dY is staged once and X once with a one-pixel
// halo on both sides (zeros outside the row, and the whole segment zero when row y + dy is
// outside the image), and the three taps read the same X tile at row offsets 0, 1, 2.  No
// per-tap masks, a third of the staging traffic, 21 MFMAs per wave and barrier pair.
// W must be a multiple of 112.  Pixel slices are runs of segments; partial tiles are added to
// the zeroed dW with float atomics.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
wgrad_rows64_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ dW, int B, int H,
                    int W, int Cin, int Cout, int splitk, int total_work, long long det_stride) {
  constexpr int SEG = 112, PX = 64 + 32, PY = 64 + 32;
  __shared__ __attribute__((aligned(16))) bf16 Xs[(SEG + 2) * PX];
  __shared__ __attribute__((aligned(16))) bf16 Ys[SEG * PY];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int per_xcd = (total_work + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_work) return;
  const int ci_tiles = Cin / 64, co_tiles = Cout / 64, tiles = 3 * ci_tiles * co_tiles;
  const int slice = work / tiles;
  int rest = work - slice * tiles;
  const int dyi = rest / (ci_tiles * co_tiles);            // filter row 0..2  (dy = dyi - 1)
  rest -= dyi * ci_tiles * co_tiles;
  const int ci0 = (rest / co_tiles) * 64, co0 = (rest % co_tiles) * 64;
  const int spr = W / SEG;                                   // segments per image row
  const int S = B * H * spr;
  const int chunk = (S + splitk - 1) / splitk;
  const int s_begin = slice * chunk, s_end = min(S, s_begin + chunk);
  if (s_begin >= s_end) return;

  // loader: two passes of 64 rows; thread -> row (tid >> 2) + 64 * pass, quarter tid & 3 of the
  // 64 channels.  Rows that do not exist read pixel 0 and are zeroed by mask on the way into LDS.
  const int l_row = tid >> 2, l_q = tid & 3;
  const bf16* x_col = X + ci0 + l_q * 16;
  const bf16* y_col = dY + co0 + l_q * 16;
  f32x16 acc[3];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  uint4 rx[2][2], ry[2][2];
  uint32_t keep_x[2] = {0u, 0u}, keep_y[2] = {0u, 0u};

#define SLAMHIP_WROWS_ISSUE(seg)                                                                    \
  {                                                                                                 \
    const int R = (seg) / spr;                       /* global image row b * H + y */               \
    const int x0 = ((seg) - R * spr) * SEG;                                                         \
    const int b = R / H, ys = R - b * H + dyi - 1;                                                  \
    const bool row_ok = (unsigned)ys < (unsigned)H;                                                 \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                 \
      const int j = l_row + 64 * i;                    /* X tile row: pixel x0 - 1 + j */           \
      const int x = x0 - 1 + j;                                                                     \
      const bool okx = j < SEG + 2 && row_ok && (unsigned)x < (unsigned)W;                          \
      const long long px = okx ? ((long long)(b * H + ys) * W + x) : 0;                             \
      const uint4* xs = reinterpret_cast<const uint4*>(x_col + px * Cin);                           \
      rx[i][0] = xs[0]; rx[i][1] = xs[1];                                                           \
      keep_x[i] = okx ? 0xffffffffu : 0u;                                                           \
      const bool oky = j < SEG;                         /* dY tile row: pixel x0 + j (< W) */       \
      const long long py = oky ? ((long long)R * W + x0 + j) : 0;                                   \
      const uint4* ysrc = reinterpret_cast<const uint4*>(y_col + py * Cout);                        \
      ry[i][0] = ysrc[0]; ry[i][1] = ysrc[1];                                                       \
      keep_y[i] = oky ? 0xffffffffu : 0u;                                                           \
    }                                                                                               \
  }

  SLAMHIP_WROWS_ISSUE(s_begin);
  for (int seg = s_begin; seg < s_end; ++seg) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int j = l_row + 64 * i;
      if (j < SEG + 2) {
        uint4* xd = reinterpret_cast<uint4*>(&Xs[j * PX + l_q * 16]);
        xd[0] = and4(rx[i][0], keep_x[i]); xd[1] = and4(rx[i][1], keep_x[i]);
      }
      if (j < SEG) {
        uint4* yd = reinterpret_cast<uint4*>(&Ys[j * PY + l_q * 16]);
        yd[0] = and4(ry[i][0], keep_y[i]); yd[1] = and4(ry[i][1], keep_y[i]);
      }
    }
    __syncthreads();
    if (seg + 1 < s_end) SLAMHIP_WROWS_ISSUE(seg + 1);
#pragma unroll
    for (int kk = 0; kk < SEG; kk += 16) {
      const bf16x8 bfr = tr_frag<PY>(Ys, kk, wn * 32, lane);              // B[k = pixel][n = co]
#pragma unroll
      for (int t = 0; t < 3; ++t) {                                        // tap dx = t - 1: X pixel k + dx = tile row k + t
        const bf16x8 af = tr_frag<PX>(Xs, kk + t, wm * 32, lane);         // A[m = ci][k = pixel]
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t], 0, 0, 0);
      }
    }
  }
#undef SLAMHIP_WROWS_ISSUE
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      const int co = co0 + wn * 32 + (lane & 31);
      float* dst = &dW[((long long)(dyi * 3 + t) * Cin + ci) * Cout + co];
      if (det_stride) dst[slice * det_stride] = acc[t][r];   // fixed-order mode: this slice's own slot
      else atomicAdd(dst, acc[t][r]);
    }
}

// ---------------------------------------------------------------------------
// The same filter-row scheme for narrow images (W + 2 <= 128): a K-step is a GROUP of G whole
// image rows, laid out in LDS with a pitch of W + 2 entries per row (dY: W pixels then two
// zeros; X: zero, W pixels, zero), G (W + 2) <= 128, the rest of the 128 entries zero.  With
// that pitch the X entry of tap dx for dY entry k is entry k + dx + 1 in every row of the group,
// and the zero halo entries stand for the pixels left and right of the image.  112 of the 128
// K entries carry pixels at W = 56, 28 and 14.  A workgroup owns one filter row of a TM x 64
// (ci x co) tile.
// ---------------------------------------------------------------------------
template <int TM>
__global__ void __launch_bounds__(256)
wgrad_rowgroup_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ dW, int BH, int H,
                      int W, int Cin, int Cout, int G, int splitk, int total_work, long long det_stride) {
  constexpr int KS = 128, PX = TM + 32, PY = 64 + 32, MT = TM / 64;
  constexpr int XT = TM / 16;                    // threads per X row (16 channels each)
  constexpr int XR = 256 / XT;                   // X rows per loader pass
  constexpr int XPASS = (KS + 2 + XR - 1) / XR;
  __shared__ __attribute__((aligned(16))) bf16 Xs[(KS + 2) * PX];
  __shared__ __attribute__((aligned(16))) bf16 Ys[KS * PY];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;

  const int per_xcd = (total_work + 7) >> 3;
  const int work = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
  if (work >= total_work) return;
  const int ci_tiles = Cin / TM, co_tiles = Cout / 64, tiles = 3 * ci_tiles * co_tiles;
  const int slice = work / tiles;
  int rest = work - slice * tiles;
  const int dyi = rest / (ci_tiles * co_tiles);            // filter row 0..2  (dy = dyi - 1)
  rest -= dyi * ci_tiles * co_tiles;
  const int ci0 = (rest / co_tiles) * TM, co0 = (rest % co_tiles) * 64;
  const int S = (BH + G - 1) / G;                            // K-steps: groups of G image rows
  const int chunk = (S + splitk - 1) / splitk;
  const int s_begin = slice * chunk, s_end = min(S, s_begin + chunk);
  if (s_begin >= s_end) return;

  // loader geometry, fixed per thread: tile entry -> (row of the group, pixel of the row)
  const int RS = W + 2;
  const int x_row = tid / XT, x_q = tid % XT;
  int x_grp[XPASS], x_off[XPASS], x_y[XPASS];
  bool x_in[XPASS];
#pragma unroll
  for (int i = 0; i < XPASS; ++i) {
    const int j = x_row + XR * i, grp = j / RS, px = j - grp * RS - 1;
    x_grp[i] = grp;
    x_off[i] = grp * W + px;
    x_in[i] = j < G * RS && (unsigned)px < (unsigned)W;
    x_y[i] = (s_begin * G + grp) % H;               // image row of this entry within its image
  }
  const int y_row = tid >> 2, y_q = tid & 3;
  int y_grp[2], y_off[2];
  bool y_in[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int j = y_row + 64 * i, grp = j / RS, px = j - grp * RS;
    y_grp[i] = grp;
    y_off[i] = grp * W + px;
    y_in[i] = j < G * RS && px < W;
  }
  const bf16* x_col = X + ci0 + x_q * 16;
  const bf16* y_col = dY + co0 + y_q * 16;
  const int g_mod = G % H;

  f32x16 acc[3][MT];
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][i][r] = 0.0f;
  uint4 rx[XPASS][2], ry[2][2];
  uint32_t keep_x[XPASS], keep_y[2];

  // rows that do not exist read pixel 0 and are zeroed by mask on the way into LDS
#define SLAMHIP_WGROUP_ISSUE(step)                                                                  \
  {                                                                                                 \
    const int R0 = (step) * G;                                                                      \
    _Pragma("unroll") for (int i = 0; i < XPASS; ++i) {                                             \
      const int ys = x_y[i] + dyi - 1;                                                              \
      const bool ok = x_in[i] && R0 + x_grp[i] < BH && (unsigned)ys < (unsigned)H;                  \
      const long long px = ok ? ((long long)(R0 + dyi - 1) * W + x_off[i]) : 0;                     \
      const uint4* xs = reinterpret_cast<const uint4*>(x_col + px * Cin);                           \
      rx[i][0] = xs[0]; rx[i][1] = xs[1];                                                           \
      keep_x[i] = ok ? 0xffffffffu : 0u;                                                            \
      x_y[i] += g_mod;                                                                              \
      if (x_y[i] >= H) x_y[i] -= H;                                                                 \
    }                                                                                               \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                 \
      const bool ok = y_in[i] && R0 + y_grp[i] < BH;                                                \
      const long long py = ok ? ((long long)R0 * W + y_off[i]) : 0;                                 \
      const uint4* ysrc = reinterpret_cast<const uint4*>(y_col + py * Cout);                        \
      ry[i][0] = ysrc[0]; ry[i][1] = ysrc[1];                                                       \
      keep_y[i] = ok ? 0xffffffffu : 0u;                                                            \
    }                                                                                               \
  }

  SLAMHIP_WGROUP_ISSUE(s_begin);
  for (int step = s_begin; step < s_end; ++step) {
    __syncthreads();
#pragma unroll
    for (int i = 0; i < XPASS; ++i) {
      const int j = x_row + XR * i;
      if (j < KS + 2) {
        uint4* xd = reinterpret_cast<uint4*>(&Xs[j * PX + x_q * 16]);
        xd[0] = and4(rx[i][0], keep_x[i]); xd[1] = and4(rx[i][1], keep_x[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      uint4* yd = reinterpret_cast<uint4*>(&Ys[(y_row + 64 * i) * PY + y_q * 16]);
      yd[0] = and4(ry[i][0], keep_y[i]); yd[1] = and4(ry[i][1], keep_y[i]);
    }
    __syncthreads();
    if (step + 1 < s_end) SLAMHIP_WGROUP_ISSUE(step + 1);
#pragma unroll
    for (int kk = 0; kk < KS; kk += 16) {
      const bf16x8 bfr = tr_frag<PY>(Ys, kk, wn * 32, lane);              // B[k = entry][n = co]
#pragma unroll
      for (int t = 0; t < 3; ++t)                                          // tap dx = t - 1: X entry k + t
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const bf16x8 af = tr_frag<PX>(Xs, kk + t, wm * (TM / 2) + i * 32, lane);   // A[m = ci][k = entry]
          acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[t][i], 0, 0, 0);
        }
    }
  }
#undef SLAMHIP_WGROUP_ISSUE
#pragma unroll
  for (int t = 0; t < 3; ++t)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + wm * (TM / 2) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const int co = co0 + wn * 32 + (lane & 31);
        float* dst = &dW[((long long)(dyi * 3 + t) * Cin + ci) * Cout + co];
        if (det_stride) dst[slice * det_stride] = acc[t][i][r];   // fixed-order mode: this slice's own slot
        else atomicAdd(dst, acc[t][i][r]);
      }
}

// ---------------------------------------------------------------------------
// first layer (Cin = 4, K = 36): too thin for the matrix cores, 0.8 % of the FLOPs
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
conv3x3_small_fwd_kernel(const bf16* __restrict__ X, const float* __restrict__ Wt /*[3,3,Cin,Cout]*/,
                         bf16* __restrict__ out, int P, int H, int W, int Cin, int Cout) {
  const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;   // over P * Cout
  if (idx >= (long long)P * Cout) return;
  const int co = (int)(idx % Cout);
  const int p = (int)(idx / Cout);
  const int hw = H * W, b = p / hw, rem = p - b * hw, y = rem / W, x = rem - y * W;
  float acc = 0.0f;
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    const bf16* xp = X + ((long long)(b * H + yy) * W + xx) * Cin;
    for (int c = 0; c < Cin; ++c) acc = __builtin_fmaf((float)xp[c], Wt[(t * Cin + c) * Cout + co], acc);
  }
  out[idx] = (bf16)acc;
}

__global__ void __launch_bounds__(256)
conv3x3_small_wgrad_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ dW, int P,
                           int H, int W, int Cin, int Cout, long long det_stride) {
  // one block per (tap, ci); thread = co; loops over a slice of the pixels
  const int t = blockIdx.x / Cin, ci = blockIdx.x % Cin;
  const int co = threadIdx.x % Cout, lanes_per_co = blockDim.x / Cout, sub = threadIdx.x / Cout;
  const int slices = gridDim.y * lanes_per_co, slice = blockIdx.y * lanes_per_co + sub;
  const int chunk = (P + slices - 1) / slices, p0 = slice * chunk, p1 = min(P, p0 + chunk);
  const int hw = H * W;
  float acc = 0.0f;
  for (int p = p0; p < p1; ++p) {
    const int b = p / hw, rem = p - b * hw, y = rem / W, x = rem - y * W;
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
    acc = __builtin_fmaf((float)X[((long long)(b * H + yy) * W + xx) * Cin + ci], (float)dY[(long long)p * Cout + co], acc);
  }
  if (det_stride) dW[slice * det_stride + (t * Cin + ci) * Cout + co] = acc;   // fixed-order mode: one slot per slice
  else atomicAdd(&dW[(t * Cin + ci) * Cout + co], acc);
}

// Specialisation for the network's actual first layer (4 input channels, 64 filters) on the
// matrix cores: the im2col row of a pixel (9 taps x 4 channels = 36 values, padded to 64) is
// built in LDS from nine 8-byte loads, so the layer is a [pixels x 64] x [64 x 64] product
// (forward) or its transpose (filter gradient) and runs at memory speed.
__device__ __forceinline__ uint2 and2(uint2 v, uint32_t keep) { return make_uint2(v.x & keep, v.y & keep); }

// forward: 128 pixels x 64 filters per workgroup, one K-step
//   MODE 0: out = z (bfloat16), column statistics of z when stat_part is given
//   MODE 1: the column statistics alone, nothing is stored (first of two passes when z is not kept)
//   MODE 2: out = y = relu(batch-norm(z)) with the statistics in `sums` (second pass): z is rounded to bfloat16
//           and normalised with the operations of bn_relu_fwd_kernel, so y equals what that kernel makes of a
//           stored z, bit for bit; the layer costs one write of y instead of a write and a read of z plus y
template <int MODE>
__global__ void __launch_bounds__(256)
conv3x3_c4_k64_fwd_kernel(const bf16* __restrict__ X, const float* __restrict__ Wt, bf16* __restrict__ out, int P,
                          int H, int W, float* __restrict__ stat_part, int tiles, const float* __restrict__ sums,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float inv_count, float eps) {
  __shared__ __attribute__((aligned(16))) bf16 As[128 * LDS_PITCH];
  __shared__ __attribute__((aligned(16))) bf16 Bs[64 * LDS_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  // B[n][k] = filter[k][n] (k = tap * 4 + channel), zero for k >= 36; once per workgroup, which
  // then walks pixel tiles blockIdx.x, blockIdx.x + gridDim.x, ...
  {
    const int n = tid & 63, kg = tid >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = kg * 16 + i;
      Bs[n * LDS_PITCH + k] = (bf16)(k < 36 ? Wt[k * 64 + n] : 0.0f);
    }
  }
  // A[pixel][k]: thread -> pixel tid>>1; half 0 fills taps 0..4, half 1 taps 5..8 and the zero tail.
  // The nine 8-byte pixels of the NEXT tile are fetched while the current one is computed and stored.
  const int row = tid >> 1, half = tid & 1;
  const uint2* Xp = reinterpret_cast<const uint2*>(X);   // one pixel = 4 bf16 = 8 bytes
  uint2 v[5];
  uint32_t keep[5];
  float bn_mean = 0.f, bn_r = 0.f, bn_beta = 0.f;        // MODE 2: a lane's accumulators all belong to one channel
  if (MODE == 2) {
    const int c = wn * 32 + (lane & 31);
    bn_mean = sums[c] * inv_count;
    const float var = fmaxf(sums[64 + c] * inv_count - bn_mean * bn_mean, 0.0f);
    bn_r = gamma[c] * rsqrtf(var + eps);
    bn_beta = beta[c];
  }
#define SLAMHIP_C4_FETCH(tile)                                                                              \
  {                                                                                                         \
    const int m0f = (tile) * 128;                                                                           \
    const bool row_ok = m0f + row < P;                                                                      \
    const int pm = row_ok ? m0f + row : P - 1;                                                              \
    const int rem = pm % (H * W), y = rem / W, x = rem - y * W;                                             \
    _Pragma("unroll") for (int i = 0; i < 5; ++i) {                                                         \
      const int t = half * 5 + i;             /* taps 0..4 | 5..9 (9 does not exist) */                     \
      const int dy = t / 3 - 1, dx = t % 3 - 1;                                                             \
      const bool ok = row_ok && t < 9 && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W; \
      v[i] = Xp[ok ? pm + dy * W + dx : pm];                                                                \
      keep[i] = ok ? 0xffffffffu : 0u;                                                                      \
    }                                                                                                       \
  }
  int tile = blockIdx.x;
  if (tile < tiles) SLAMHIP_C4_FETCH(tile);
  for (; tile < tiles; tile += gridDim.x) {
    const int m0 = tile * 128;
    __syncthreads();          // the previous tile's staged output has been read (and Bs is written)
    {
      uint2* dst = reinterpret_cast<uint2*>(&As[row * LDS_PITCH]);
#pragma unroll
      for (int i = 0; i < 5; ++i) dst[half * 5 + i] = and2(v[i], keep[i]);   // k = 4 * tap ... (tap 9 = zeros)
      if (half) {
#pragma unroll
        for (int i = 10; i < 16; ++i) dst[i] = make_uint2(0u, 0u);           // k = 40..63
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < tiles) SLAMHIP_C4_FETCH(tile + (int)gridDim.x);
    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 48; kk += 16) {   // k >= 48 is all zeros
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 32 + (lane & 31)) * LDS_PITCH + kk + 8 * (lane >> 5)]);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(&As[(wm * 64 + i * 32 + (lane & 31)) * LDS_PITCH + kk + 8 * (lane >> 5)]);
        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, acc[i], 0, 0, 0);
      }
    }
    // the output tile goes through LDS (the im2col rows are no longer needed) so that it leaves
    // in 16-byte stores of whole 128-byte pixel rows instead of 2-byte stores
    if (MODE != 1) {
      __syncthreads();
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          bf16 o = (bf16)acc[i][r];
          if (MODE == 2) o = (bf16)fmaxf(__builtin_fmaf((float)o - bn_mean, bn_r, bn_beta), 0.0f);
          As[(wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * LDS_PITCH + wn * 32 + (lane & 31)] = o;
        }
      __syncthreads();
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int id = tid + 256 * k, orow = id >> 3, c8 = id & 7;
        if (m0 + orow < P)
          *reinterpret_cast<uint4*>(out + (long long)(m0 + orow) * 64 + c8 * 8) = *reinterpret_cast<const uint4*>(&As[orow * LDS_PITCH + c8 * 8]);
      }
    }
    if (MODE != 2 && stat_part) {
      float sm = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float vv = (float)(bf16)acc[i][r];
          sm += vv;
          sq = __builtin_fmaf(vv, vv, sq);
        }
      store_column_stats(sm, sq, stat_part + ((long long)tile * 2 + wm) * 128, wn * 32 + (lane & 31), 64, lane);
    }
  }
#undef SLAMHIP_C4_FETCH
}

// filter gradient: dW[k][n] += sum_p A[p][k] dY[p][n] over a slice of the pixels
// FUSE_BN: dY is not read; it is computed on the way into LDS from the gradient g at the
// layer's output and z, i.e. ReLU gate and batch-norm backward of bn_relu_bwd_apply_kernel with
// the same operations and the same rounding to bfloat16, so that dz of the first layer is
// neither written nor read (its only reader is this kernel).
template <bool FUSE_BN>
__global__ void __launch_bounds__(256)
conv3x3_c4_k64_wgrad_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, float* __restrict__ dW, int P,
                            int H, int W, int px_per_block, const bf16* __restrict__ Z,
                            const float* __restrict__ sums, const float* __restrict__ red,
                            const float* __restrict__ gamma, const float* __restrict__ beta, float inv_count,
                            float eps, long long det_stride) {
  constexpr int PX = 64 + 32;   // conflict-free pitch of the transposing reads (see wgrad_bf16_kernel)
  __shared__ __attribute__((aligned(16))) bf16 Xs[64 * PX];
  __shared__ __attribute__((aligned(16))) bf16 Ys[64 * PX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const long long p_begin_l = (long long)blockIdx.x * px_per_block;
  if (p_begin_l >= P) return;
  const int p_begin = (int)p_begin_l, p_end = min(P, p_begin + px_per_block);
  const int l_row = tid >> 2, l_q = tid & 3;   // pixel of the 64-pixel step, quarter of its row
  const uint2* Xp = reinterpret_cast<const uint2*>(X);
  const int hw = H * W;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  uint2 xv[3];
  uint4 yv[2], zv[2];
  uint32_t keep_x[3], keep_y = 0u;
  // FUSE_BN: coefficients of this thread's 16 channels (l_q * 16 ...), as in bn_relu_bwd_apply_kernel
  float mean[16], rr[16], bet[16], k1[16], k2[16];
  if (FUSE_BN) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = l_q * 16 + j;
      mean[j] = sums[c] * inv_count;
      const float var = fmaxf(sums[64 + c] * inv_count - mean[j] * mean[j], 0.0f);
      const float rstd = rsqrtf(var + eps);
      rr[j] = gamma[c] * rstd;
      bet[j] = beta[c];
      k1[j] = red[c] * inv_count;
      k2[j] = rstd * (red[64 + c] * inv_count);
    }
  }

#define SLAMHIP_C4_ISSUE(p0)                                                                    \
  {                                                                                             \
    const int p = (p0) + l_row;                                                                 \
    const bool in_slice = p < p_end;                                                            \
    const int pc = in_slice ? p : p_end - 1;                                                    \
    const int rem = pc % hw, y = rem / W, x = rem - y * W;                                      \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
      const int t = l_q * 3 + i;                        /* quarter 3 owns no tap: zeros */      \
      const int dy = t / 3 - 1, dx = t % 3 - 1;                                                 \
      const bool ok = in_slice && t < 9 && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W; \
      xv[i] = Xp[ok ? pc + dy * W + dx : pc];                                                   \
      keep_x[i] = ok ? 0xffffffffu : 0u;                                                        \
    }                                                                                           \
    const uint4* ys = reinterpret_cast<const uint4*>(dY + (long long)pc * 64 + l_q * 16);       \
    yv[0] = ys[0]; yv[1] = ys[1];                                                               \
    if (FUSE_BN) {                                                                              \
      const uint4* zs = reinterpret_cast<const uint4*>(Z + (long long)pc * 64 + l_q * 16);      \
      zv[0] = zs[0]; zv[1] = zs[1];                                                             \
    }                                                                                           \
    keep_y = in_slice ? 0xffffffffu : 0u;                                                       \
  }

  SLAMHIP_C4_ISSUE(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    __syncthreads();
    {
      uint2* xd = reinterpret_cast<uint2*>(&Xs[l_row * PX]);
      // quarters 0..2 write taps 3q..3q+2 (k = 12q ... 12q+11); quarter 3 the zero tail k = 36..63
      if (l_q < 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) xd[l_q * 3 + i] = and2(xv[i], keep_x[i]);
      } else {
#pragma unroll
        for (int i = 9; i < 16; ++i) xd[i] = make_uint2(0u, 0u);
      }
      uint4* yd = reinterpret_cast<uint4*>(&Ys[l_row * PX + l_q * 16]);
      if (FUSE_BN) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const bf16x8 vg = __builtin_bit_cast(bf16x8, yv[h]), vz = __builtin_bit_cast(bf16x8, zv[h]);
          bf16x8 o;
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const int c = h * 8 + j;
            const float d = (float)vz[j] - mean[c];
            const float g = (__builtin_fmaf(d, rr[c], bet[c]) > 0.0f) ? (float)vg[j] : 0.0f;   // forward's y > 0
            o[j] = (bf16)(rr[c] * ((g - k1[c]) - d * k2[c]));
          }
          yd[h] = and4(__builtin_bit_cast(uint4, o), keep_y);
        }
      } else {
        yd[0] = and4(yv[0], keep_y); yd[1] = and4(yv[1], keep_y);
      }
    }
    __syncthreads();
    if (p0 + 64 < p_end) SLAMHIP_C4_ISSUE(p0 + 64);
#pragma unroll
    for (int kk = 0; kk < 64; kk += 16) {
      const bf16x8 af = tr_frag<PX>(Xs, kk, wm * 32, lane);   // A[m = k-index][k = pixel]
      const bf16x8 bf = tr_frag<PX>(Ys, kk, wn * 32, lane);   // B[k = pixel][n = filter]
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
    }
  }
#undef SLAMHIP_C4_ISSUE
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (k < 36) {
      float* dst = &dW[k * 64 + wn * 32 + (lane & 31)];
      if (det_stride) dst[blockIdx.x * det_stride] = acc[r];   // fixed-order mode: one slot per workgroup
      else atomicAdd(dst, acc[r]);
    }
  }
}

// Backward of the first layer WITHOUT its z tensor.  The filter gradient stages the im2col rows of every pixel
// anyway, so z is recomputed from them per 64-pixel step: three MFMAs per wave with the forward kernel's operands
// and order (pixels as A, the bfloat16 filter as B, k = 0, 16, 32), i.e. the forward's values.  In the accumulator
// layout a lane's sixteen values belong to ONE channel, so the gate and the batch-norm backward need one set of
// coefficients per lane; dy is staged in LDS as it comes from memory (16-byte loads) and every lane reads, and in
// PHASE 1 overwrites with dz, the sixteen 2-byte entries that match its accumulators.  The 822 MB of z at batch 128
// are then neither written by the forward pass nor read by the two backward passes.
//   PHASE 0: one record of 128 floats per workgroup: sum g and sum g * xhat per channel, g = dy gated by y > 0
//            (channel_reduce_kernel<1>'s quantities; the records are added by stat_finish_kernel)
//   PHASE 1: dW += A^T dz with dz = batch-norm backward of g (bn_relu_bwd_apply_kernel's operations and rounding)
template <int PHASE>
__global__ void __launch_bounds__(256)
conv3x3_c4_k64_bwd_kernel(const bf16* __restrict__ X, const bf16* __restrict__ dY, const float* __restrict__ Wt,
                          float* __restrict__ dW, float* __restrict__ red_part, int P, int H, int W,
                          int px_per_block, const float* __restrict__ sums, const float* __restrict__ red,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float inv_count,
                          float eps, long long det_stride) {
  constexpr int PX = 64 + 32;   // conflict-free pitch of the transposing reads (see wgrad_bf16_kernel)
  __shared__ __attribute__((aligned(16))) bf16 Xs[64 * PX];          // im2col rows [pixel][k]
  __shared__ __attribute__((aligned(16))) bf16 Ys[64 * PX];          // dy, then dz [pixel][channel]
  __shared__ __attribute__((aligned(16))) bf16 Bs[64 * LDS_PITCH];   // filter [channel][k]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;     // wave: pixels wm * 32 ..., channels wn * 32 ... of the step
  const long long p_begin_l = (long long)blockIdx.x * px_per_block;
  if (p_begin_l >= P) return;
  const int p_begin = (int)p_begin_l, p_end = min(P, p_begin + px_per_block);
  const int l_row = tid >> 2, l_q = tid & 3;   // loader: pixel of the 64-pixel step, quarter of its row
  const uint2* Xp = reinterpret_cast<const uint2*>(X);
  const int hw = H * W;
  {   // B[n][k] = filter[k][n] (k = tap * 4 + channel), zero for k >= 36, as the forward kernel builds it
    const int n = tid & 63, kg = tid >> 6;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = kg * 16 + i;
      Bs[n * LDS_PITCH + k] = (bf16)(k < 36 ? Wt[k * 64 + n] : 0.0f);
    }
  }
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  uint2 xv[3];
  uint4 yv[2];
  uint32_t keep_x[3], keep_y = 0u;
  // this lane's channel: column wn * 32 + (lane & 31) of the z tile
  const int c = wn * 32 + (lane & 31);
  const float mean = sums[c] * inv_count;
  const float rstd = rsqrtf(fmaxf(sums[64 + c] * inv_count - mean * mean, 0.0f) + eps);
  const float rr = gamma[c] * rstd, bet = beta[c];
  const float k1 = PHASE == 1 ? red[c] * inv_count : 0.f;
  const float k2 = PHASE == 1 ? rstd * (red[64 + c] * inv_count) : 0.f;
  float s0 = 0.f, s1 = 0.f;
  // entry of accumulator r in the dy / dz tile: row wm * 32 + (r & 3) + 8 (r >> 2) + 4 (lane >> 5), column c
  bf16* const my_y = &Ys[(wm * 32 + 4 * (lane >> 5)) * PX + c];

#define SLAMHIP_C4B_ISSUE(p0)                                                                   \
  {                                                                                             \
    const int p = (p0) + l_row;                                                                 \
    const bool in_slice = p < p_end;                                                            \
    const int pc = in_slice ? p : p_end - 1;                                                    \
    const int rem = pc % hw, y = rem / W, x = rem - y * W;                                      \
    _Pragma("unroll") for (int i = 0; i < 3; ++i) {                                             \
      const int t = l_q * 3 + i;                        /* quarter 3 owns no tap: zeros */      \
      const int dy = t / 3 - 1, dx = t % 3 - 1;                                                 \
      const bool ok = in_slice && t < 9 && (unsigned)(y + dy) < (unsigned)H && (unsigned)(x + dx) < (unsigned)W; \
      xv[i] = Xp[ok ? pc + dy * W + dx : pc];                                                   \
      keep_x[i] = ok ? 0xffffffffu : 0u;                                                        \
    }                                                                                           \
    const uint4* ys = reinterpret_cast<const uint4*>(dY + (long long)pc * 64 + l_q * 16);       \
    yv[0] = ys[0]; yv[1] = ys[1];                                                               \
    keep_y = in_slice ? 0xffffffffu : 0u;                                                       \
  }

  SLAMHIP_C4B_ISSUE(p_begin);
  for (int p0 = p_begin; p0 < p_end; p0 += 64) {
    __syncthreads();          // the previous step's reads of Xs and Ys are done (and Bs is written)
    {
      uint2* xd = reinterpret_cast<uint2*>(&Xs[l_row * PX]);
      // quarters 0..2 write taps 3q..3q+2 (k = 12q ... 12q+11); quarter 3 the zero tail k = 36..63
      if (l_q < 3) {
#pragma unroll
        for (int i = 0; i < 3; ++i) xd[l_q * 3 + i] = and2(xv[i], keep_x[i]);
      } else {
#pragma unroll
        for (int i = 9; i < 16; ++i) xd[i] = make_uint2(0u, 0u);
      }
      uint4* yd = reinterpret_cast<uint4*>(&Ys[l_row * PX + l_q * 16]);     // rows beyond the slice: dy = 0
      yd[0] = and4(yv[0], keep_y); yd[1] = and4(yv[1], keep_y);
    }
    __syncthreads();
    if (p0 + 64 < p_end) SLAMHIP_C4B_ISSUE(p0 + 64);
    f32x16 zacc;
#pragma unroll
    for (int r = 0; r < 16; ++r) zacc[r] = 0.0f;
#pragma unroll
    for (int kk = 0; kk < 48; kk += 16) {   // k >= 48 is all zeros
      const bf16x8 bfr = *reinterpret_cast<const bf16x8*>(&Bs[(wn * 32 + (lane & 31)) * LDS_PITCH + kk + 8 * (lane >> 5)]);
      const bf16x8 af = *reinterpret_cast<const bf16x8*>(&Xs[(wm * 32 + (lane & 31)) * PX + kk + 8 * (lane >> 5)]);
      zacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bfr, zacc, 0, 0, 0);
    }
    const bool rows_in = p0 + wm * 32 + 32 <= p_end;   // PHASE 1: dz of rows beyond the slice must stay zero
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      bf16* const e = my_y + ((r & 3) + 8 * (r >> 2)) * PX;
      const float d = (float)(bf16)zacc[r] - mean;
      const float g = (__builtin_fmaf(d, rr, bet) > 0.0f) ? (float)*e : 0.0f;   // forward's y > 0
      if (PHASE == 1) {
        const bool in = rows_in || p0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) < p_end;
        *e = in ? (bf16)(rr * ((g - k1) - d * k2)) : (bf16)0.0f;
      } else {
        s0 += g;
        s1 = __builtin_fmaf(g, d, s1);      // times rstd at the end
      }
    }
    if (PHASE == 1) {
      __syncthreads();
#pragma unroll
      for (int kk = 0; kk < 64; kk += 16) {
        const bf16x8 af = tr_frag<PX>(Xs, kk, wm * 32, lane);   // A[m = k-index][k = pixel]
        const bf16x8 bf = tr_frag<PX>(Ys, kk, wn * 32, lane);   // B[k = pixel][n = filter]
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, acc, 0, 0, 0);
      }
    }
  }
#undef SLAMHIP_C4B_ISSUE
  if (PHASE == 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      if (k < 36) {
        float* dst = &dW[k * 64 + wn * 32 + (lane & 31)];
        if (det_stride) dst[blockIdx.x * det_stride] = acc[r];   // fixed-order mode: one slot per workgroup
        else atomicAdd(dst, acc[r]);
      }
    }
  } else {
    // lanes l and l + 32 hold the same channel; the two pixel halves (waves wm = 0, 1) meet in LDS, in a fixed order
    s1 *= rstd;                                         // sum g d -> sum g xhat
    s0 += __shfl_xor(s0, 32);
    s1 += __shfl_xor(s1, 32);
    __syncthreads();                                    // the last step's reads of Ys are done
    float* part = reinterpret_cast<float*>(Ys);         // [wm][which][64 channels]
    if (lane < 32) {
      part[(wm * 2 + 0) * 64 + c] = s0;
      part[(wm * 2 + 1) * 64 + c] = s1;
    }
    __syncthreads();
    if (tid < 128)            // record[c] = sum g, record[64 + c] = sum g xhat
      red_part[(long long)blockIdx.x * 128 + tid] = part[tid] + part[128 + tid];
  }
}

// ---------------------------------------------------------------------------
// per-channel statistics and batch-norm + ReLU (cnn_model.py:104-113; batch
// statistics always, biased variance)
// ---------------------------------------------------------------------------
// Column reductions over the P rows of an [P][C] bfloat16 matrix, 16 bytes (8 channels) per
// thread per row: a block is (C/8 column groups) x (256 / (C/8) row lanes), rows strided over
// the grid; partial sums meet in LDS and leave as 2C float atomics per block.
//   BWD = 0: out[c] += sum z,   out[C + c] += sum z^2                        (statistics)
//   BWD = 1: out[c] += sum g,   out[C + c] += sum g * xhat,  g = dy * (y > 0) (batch-norm backward)
//   BWD = 2: out[c] += sum g                                                  (bias backward; out has C entries)
template <int BWD>
__global__ void __launch_bounds__(256)
channel_reduce_kernel(const bf16* __restrict__ a, const bf16* __restrict__ y, const bf16* __restrict__ z,
                      const float* __restrict__ sums, const float* __restrict__ gamma,
                      const float* __restrict__ beta, float* __restrict__ out, int P, int C, float inv_count,
                      float eps, int det) {
  __shared__ float part[256 * 17];
  const int groups = C >> 3;                      // <= 256 (checked on the host)
  const int rlanes = 256 / groups;
  const int gcol = threadIdx.x % groups, rlane = threadIdx.x / groups;
  float s0[8], s1[8], mean[8], rstd[8], gam[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; mean[j] = 0.f; rstd[j] = 0.f; gam[j] = 0.f; bet[j] = 0.f; }
  if (rlane < rlanes) {
    if (BWD == 1) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int c = gcol * 8 + j;
        mean[j] = sums[c] * inv_count;
        rstd[j] = rsqrtf(fmaxf(sums[C + c] * inv_count - mean[j] * mean[j], 0.0f) + eps);
        gam[j] = gamma[c] * rstd[j];     // the forward pass's scale, same operations
        bet[j] = beta[c];
      }
    }
    const int stride = gridDim.x * rlanes;
    for (int p = blockIdx.x * rlanes + rlane; p < P; p += stride) {
      const long long o = ((long long)p * C >> 3) + gcol;
      const bf16x8 va = reinterpret_cast<const bf16x8*>(a)[o];
      if (BWD == 1) {
        // the ReLU gate is recomputed from z exactly as the forward pass computed y: y is not read
        const bf16x8 vz = reinterpret_cast<const bf16x8*>(z)[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float d = (float)vz[j] - mean[j];
          const float g = (__builtin_fmaf(d, gam[j], bet[j]) > 0.0f) ? (float)va[j] : 0.0f;
          s0[j] += g;
          s1[j] = __builtin_fmaf(g, d * rstd[j], s1[j]);
        }
      } else if (BWD == 2) {
        const bf16x8 vy = reinterpret_cast<const bf16x8*>(y)[o];
#pragma unroll
        for (int j = 0; j < 8; ++j) s0[j] += ((float)vy[j] > 0.0f) ? (float)va[j] : 0.0f;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float v = (float)va[j];
          s0[j] += v;
          s1[j] = __builtin_fmaf(v, v, s1[j]);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { part[threadIdx.x * 17 + j] = s0[j]; part[threadIdx.x * 17 + 8 + j] = s1[j]; }
  __syncthreads();
  // thread t < 16 * groups: column group t / 16, quantity t % 16, summed over the row lanes
  for (int t = threadIdx.x; t < 16 * groups; t += 256) {
    const int gc = t >> 4, k = t & 15;
    if (BWD == 2 && k >= 8) continue;
    float acc = 0.f;
    for (int r = 0; r < rlanes; ++r) acc += part[(r * groups + gc) * 17 + k];
    float* dst = &out[(k >> 3) * C + gc * 8 + (k & 7)];
    if (det) dst[(long long)blockIdx.x * (BWD == 2 ? C : 2 * C)] = acc;   // fixed-order mode: one slot per workgroup
    else atomicAdd(dst, acc);
  }
}

// second stage of the statistics that the convolution kernels leave per tile:
// sums[c] += sum over the R records of part[r][c], c in [0, 2N)
__global__ void __launch_bounds__(256)
stat_finish_kernel(const float* __restrict__ part, float* __restrict__ sums, int R, int N2, int det) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= N2) return;
  const int chunk = (R + gridDim.y - 1) / gridDim.y, r0 = blockIdx.y * chunk, r1 = min(R, r0 + chunk);
  // eight independent loads in flight per thread: the walk is latency-bound otherwise
  float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int r = r0;
  for (; r + 7 < r1; r += 8) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = part[(long long)(r + j) * N2 + c];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += v[j];
  }
  for (; r < r1; ++r) a[0] += part[(long long)r * N2 + c];
  const float t = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
  if (det) sums[(long long)blockIdx.y * N2 + c] = t;     // fixed-order mode: one slot per slice of the records (zero if empty)
  else if (r0 < r1) atomicAdd(&sums[c], t);
}

// Both element-wise batch-norm kernels walk the tensor with a stride that is a multiple of C
// (the host picks the grid), so a thread meets the same 8 channels in every iteration and
// keeps their coefficients in registers.
// y = relu(gamma * (z - mean) * rsqrt(var + eps) + beta), 8 channels per thread
__global__ void __launch_bounds__(256)
bn_relu_fwd_kernel(const bf16* __restrict__ z, const float* __restrict__ sums, const float* __restrict__ gamma,
                   const float* __restrict__ beta, bf16* __restrict__ y, long long total8, int C, float inv_count,
                   float eps) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int c0 = (int)((i * 8) % C);
  float mean[8], r[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = sums[c0 + j] * inv_count;
    const float var = fmaxf(sums[C + c0 + j] * inv_count - mean[j] * mean[j], 0.0f);
    r[j] = gamma[c0 + j] * rsqrtf(var + eps);
    bet[j] = beta[c0 + j];
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total8; i += stride) {
    const bf16x8 v = reinterpret_cast<const bf16x8*>(z)[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(__builtin_fmaf((float)v[j] - mean[j], r[j], bet[j]), 0.0f);
    reinterpret_cast<bf16x8*>(y)[i] = o;
  }
}

// dz = gamma * rstd * (dyr - mean(dyr) - xhat * mean(dyr * xhat)), dyr = dy gated by the forward's y > 0
__global__ void __launch_bounds__(256)
bn_relu_bwd_apply_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ z,
                         const float* __restrict__ sums, const float* __restrict__ red,
                         const float* __restrict__ gamma, const float* __restrict__ beta, bf16* __restrict__ dz,
                         long long total8, int C, float inv_count, float eps) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int c0 = (int)((i * 8) % C);
  float mean[8], r[8], bet[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    mean[j] = sums[c] * inv_count;
    const float var = fmaxf(sums[C + c] * inv_count - mean[j] * mean[j], 0.0f);
    const float rstd = rsqrtf(var + eps);
    r[j] = gamma[c] * rstd;
    bet[j] = beta[c];
    k1[j] = red[c] * inv_count;
    k2[j] = rstd * (red[C + c] * inv_count);
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total8; i += stride) {
    const bf16x8 vdy = reinterpret_cast<const bf16x8*>(dy)[i], vz = reinterpret_cast<const bf16x8*>(z)[i];
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float d = (float)vz[j] - mean[j];
      const float g = (__builtin_fmaf(d, r[j], bet[j]) > 0.0f) ? (float)vdy[j] : 0.0f;   // forward's y > 0
      o[j] = (bf16)(r[j] * ((g - k1[j]) - d * k2[j]));
    }
    reinterpret_cast<bf16x8*>(dz)[i] = o;
  }
}

// ---------------------------------------------------------------------------
// bias + ReLU (cnn_model_noBN.py:109, cnn_model.py:123) and its backward
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
bias_act_fwd_kernel(const float* __restrict__ z, const float* __restrict__ bias, bf16* __restrict__ y_bf16,
                    float* __restrict__ y_f32, long long total, int N, int relu) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  float v = z[i] + bias[i % N];
  if (relu) v = fmaxf(v, 0.0f);
  if (y_bf16) y_bf16[i] = (bf16)v;
  if (y_f32) y_f32[i] = v;
}

// dz = dy * (y > 0) (relu) or dy; dbias[n] += sum_rows dz
__global__ void __launch_bounds__(256)
bias_act_bwd_kernel(const float* __restrict__ dy, const bf16* __restrict__ y, bf16* __restrict__ dz,
                    float* __restrict__ dbias, int M, int N, int relu) {
  // 64 columns x 4 row groups per workgroup: a thread walks every fourth row (loads of eight
  // rows in flight), the four partial column sums meet in LDS
  __shared__ float part[4][64];
  const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
  const int n = blockIdx.x * 64 + col;
  float s = 0.f;
  if (n < N) {
#pragma unroll 8
    for (int m = rg; m < M; m += 4) {
      const long long o = (long long)m * N + n;
      const float g = (!relu || (float)y[o] > 0.0f) ? dy[o] : 0.0f;
      dz[o] = (bf16)g;
      s += g;
    }
  }
  part[rg][col] = s;
  __syncthreads();
  if (rg == 0 && n < N) dbias[n] = (part[0][col] + part[1][col]) + (part[2][col] + part[3][col]);
}

// y = relu(z + bias) on NHWC bfloat16 (the convolution layer without batch-norm,
// cnn_model_noBN.py:107-111), and the ReLU gate of its backward pass
__global__ void __launch_bounds__(256)
bias_relu_nhwc_fwd_kernel(const bf16* __restrict__ z, const float* __restrict__ bias, bf16* __restrict__ y,
                          long long total8, int C) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const int c0 = (int)((i * 8) % C);
  const bf16x8 v = reinterpret_cast<const bf16x8*>(z)[i];
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf((float)v[j] + bias[c0 + j], 0.0f);
  reinterpret_cast<bf16x8*>(y)[i] = o;
}

__global__ void __launch_bounds__(256)
relu_gate_kernel(const bf16* __restrict__ dy, const bf16* __restrict__ y, bf16* __restrict__ dz, long long total8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total8) return;
  const bf16x8 vd = reinterpret_cast<const bf16x8*>(dy)[i], vy = reinterpret_cast<const bf16x8*>(y)[i];
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = ((float)vy[j] > 0.0f) ? vd[j] : (bf16)0.0f;
  reinterpret_cast<bf16x8*>(dz)[i] = o;
}

// ---------------------------------------------------------------------------
// 2x2 stride-2 VALID max-pool (cnn_model.py:40) and its backward (gradient to
// the first maximum in row-major window order, like TensorFlow)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
maxpool2_fwd_kernel(const bf16* __restrict__ x, bf16* __restrict__ y, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * Ho * Wo * C8) return;
  const int c8 = (int)(i % C8);
  long long r = i / C8;
  const int xo = (int)(r % Wo); r /= Wo;
  const int yo = (int)(r % Ho);
  const int b = (int)(r / Ho);
  const bf16x8* base = reinterpret_cast<const bf16x8*>(x + (((long long)b * H + 2 * yo) * W + 2 * xo) * C) + c8;
  const bf16x8 v0 = base[0], v1 = base[C8], v2 = base[(long long)W * C8], v3 = base[(long long)W * C8 + C8];
  bf16x8 o;
#pragma unroll
  for (int j = 0; j < 8; ++j) o[j] = (bf16)fmaxf(fmaxf((float)v0[j], (float)v1[j]), fmaxf((float)v2[j], (float)v3[j]));
  reinterpret_cast<bf16x8*>(y)[i] = o;
}

__global__ void __launch_bounds__(256)
maxpool2_bwd_kernel(const bf16* __restrict__ x, const bf16* __restrict__ y, const bf16* __restrict__ dy,
                    bf16* __restrict__ dx, int B, int H, int W, int C) {
  const int Ho = H / 2, Wo = W / 2, C8 = C / 8;
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * Ho * Wo * C8) return;
  const int c8 = (int)(i % C8);
  long long r = i / C8;
  const int xo = (int)(r % Wo); r /= Wo;
  const int yo = (int)(r % Ho);
  const int b = (int)(r / Ho);
  const long long o00 = ((((long long)b * H + 2 * yo) * W + 2 * xo) * C) / 8 + c8;
  const long long offs[4] = {o00, o00 + C8, o00 + (long long)W * C8, o00 + (long long)W * C8 + C8};
  const bf16x8 vy = reinterpret_cast<const bf16x8*>(y)[i], vd = reinterpret_cast<const bf16x8*>(dy)[i];
  bf16x8 in[4], out[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) in[q] = reinterpret_cast<const bf16x8*>(x)[offs[q]];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bool taken = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool hit = !taken && ((float)in[q][j] == (float)vy[j]);
      out[q][j] = hit ? vd[j] : (bf16)0.0f;
      taken = taken || hit;
    }
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) reinterpret_cast<bf16x8*>(dx)[offs[q]] = out[q];
  // odd trailing rows / columns of x (none for the 224 -> 7 chain) are zeroed by the caller
}

// ---------------------------------------------------------------------------
// batch-norm + ReLU fused with the 2x2 max-pool that follows it (the last layer of
// every block, cnn_model.py:37-62).  One thread owns a pooling window x 8 channels.
// Forward: z -> y (kept for the next block's filter gradient) and pooled in one pass.
// Backward: the pooled gradient goes to the window's first maximum of y (row-major
// order, as maxpool2_bwd_kernel) and y is recomputed from z with the forward's own
// operations, so neither y nor an upsampled gradient tensor is read or written.
// ---------------------------------------------------------------------------
struct PoolWindow {
  long long o[4];     // bf16x8 indices of the four pixels
};
__device__ __forceinline__ PoolWindow pool_window(long long pp, int c8, int H, int W, int C8) {
  const int Wo = W >> 1, Ho = H >> 1;
  const int xo = (int)(pp % Wo);
  const long long r = pp / Wo;
  const int yo = (int)(r % Ho);
  const long long b = r / Ho;
  PoolWindow w;
  w.o[0] = ((b * H + 2 * yo) * W + 2 * xo) * C8 + c8;
  w.o[1] = w.o[0] + C8;
  w.o[2] = w.o[0] + (long long)W * C8;
  w.o[3] = w.o[2] + C8;
  return w;
}

__global__ void __launch_bounds__(256)
bn_relu_pool_fwd_kernel(const bf16* __restrict__ z, const float* __restrict__ sums, const float* __restrict__ gamma,
                        const float* __restrict__ beta, bf16* __restrict__ y, bf16* __restrict__ pooled,
                        long long total8, int H, int W, int C, float inv_count, float eps) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // over pooled pixels x C/8
  if (i >= total8) return;
  const int C8 = C >> 3;
  const int c8 = (int)(i % C8), c0 = c8 * 8;
  float mean[8], r[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    mean[j] = sums[c0 + j] * inv_count;
    const float var = fmaxf(sums[C + c0 + j] * inv_count - mean[j] * mean[j], 0.0f);
    r[j] = gamma[c0 + j] * rsqrtf(var + eps);
    bet[j] = beta[c0 + j];
  }
  const long long stride = (long long)gridDim.x * blockDim.x;          // a multiple of C/8
  for (; i < total8; i += stride) {
    const PoolWindow w = pool_window(i / C8, c8, H, W, C8);
    bf16x8 v[4], best;
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const bf16x8*>(z)[w.o[q]];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      bf16x8 o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        o[j] = (bf16)fmaxf(__builtin_fmaf((float)v[q][j] - mean[j], r[j], bet[j]), 0.0f);
        best[j] = (q == 0 || (float)o[j] > (float)best[j]) ? o[j] : best[j];
      }
      if (y) reinterpret_cast<bf16x8*>(y)[w.o[q]] = o;   // optional: nothing downstream of a pooled layer reads it
    }
    reinterpret_cast<bf16x8*>(pooled)[i] = best;
  }
}

// out[c] += sum g, out[C + c] += sum g * xhat over the pixels that won their pooling window
__global__ void __launch_bounds__(256)
bn_pool_bwd_reduce_kernel(const bf16* __restrict__ dpooled, const bf16* __restrict__ z, const float* __restrict__ sums,
                          const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ out,
                          long long PP, int H, int W, int C, float inv_count, float eps, int det) {
  __shared__ float part[256 * 17];
  const int groups = C >> 3;
  const int rlanes = 256 / groups;
  const int gcol = threadIdx.x % groups, rlane = threadIdx.x / groups;
  float s0[8], s1[8], mean[8], rstd[8], gam[8], bet[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { s0[j] = 0.f; s1[j] = 0.f; mean[j] = 0.f; rstd[j] = 0.f; gam[j] = 0.f; bet[j] = 0.f; }
  if (rlane < rlanes) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = gcol * 8 + j;
      mean[j] = sums[c] * inv_count;
      rstd[j] = rsqrtf(fmaxf(sums[C + c] * inv_count - mean[j] * mean[j], 0.0f) + eps);
      gam[j] = gamma[c] * rstd[j];
      bet[j] = beta[c];
    }
    const long long stride = (long long)gridDim.x * rlanes;
    for (long long pp = (long long)blockIdx.x * rlanes + rlane; pp < PP; pp += stride) {
      const PoolWindow w = pool_window(pp, gcol, H, W, groups);
      const bf16x8 g = reinterpret_cast<const bf16x8*>(dpooled)[pp * groups + gcol];
      bf16x8 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const bf16x8*>(z)[w.o[q]];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float best = 0.0f, dbest = 0.0f;         // y of the winner and its z - mean
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d = (float)v[q][j] - mean[j];
          const float yq = (float)(bf16)fmaxf(__builtin_fmaf(d, gam[j], bet[j]), 0.0f);
          if (q == 0 || yq > best) { best = yq; dbest = d; }
        }
        const float ge = best > 0.0f ? (float)g[j] : 0.0f;
        s0[j] += ge;
        s1[j] = __builtin_fmaf(ge, dbest * rstd[j], s1[j]);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) { part[threadIdx.x * 17 + j] = s0[j]; part[threadIdx.x * 17 + 8 + j] = s1[j]; }
  __syncthreads();
  for (int t = threadIdx.x; t < 16 * groups; t += 256) {
    const int gc = t >> 4, k = t & 15;
    float acc = 0.f;
    for (int r = 0; r < rlanes; ++r) acc += part[(r * groups + gc) * 17 + k];
    float* dst = &out[(k >> 3) * C + gc * 8 + (k & 7)];
    if (det) dst[(long long)blockIdx.x * 2 * C] = acc;   // fixed-order mode: one slot per workgroup
    else atomicAdd(dst, acc);
  }
}

__global__ void __launch_bounds__(256)
bn_pool_bwd_apply_kernel(const bf16* __restrict__ dpooled, const bf16* __restrict__ z, const float* __restrict__ sums,
                         const float* __restrict__ red, const float* __restrict__ gamma,
                         const float* __restrict__ beta, bf16* __restrict__ dz, long long total8, int H, int W, int C,
                         float inv_count, float eps) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // over pooled pixels x C/8
  if (i >= total8) return;
  const int C8 = C >> 3;
  const int c8 = (int)(i % C8), c0 = c8 * 8;
  float mean[8], r[8], bet[8], k1[8], k2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = c0 + j;
    mean[j] = sums[c] * inv_count;
    const float var = fmaxf(sums[C + c] * inv_count - mean[j] * mean[j], 0.0f);
    const float rstd = rsqrtf(var + eps);
    r[j] = gamma[c] * rstd;
    bet[j] = beta[c];
    k1[j] = red[c] * inv_count;
    k2[j] = rstd * (red[C + c] * inv_count);
  }
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < total8; i += stride) {
    const PoolWindow w = pool_window(i / C8, c8, H, W, C8);
    const bf16x8 g = reinterpret_cast<const bf16x8*>(dpooled)[i];
    bf16x8 v[4], o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = reinterpret_cast<const bf16x8*>(z)[w.o[q]];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float d[4], best = 0.0f;
      int win = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        d[q] = (float)v[q][j] - mean[j];
        const float yq = (float)(bf16)fmaxf(__builtin_fmaf(d[q], r[j], bet[j]), 0.0f);
        if (q == 0 || yq > best) { best = yq; win = q; }
      }
      const float ge = best > 0.0f ? (float)g[j] : 0.0f;
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q][j] = (bf16)(r[j] * (((q == win ? ge : 0.0f) - k1[j]) - d[q] * k2[j]));
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) reinterpret_cast<bf16x8*>(dz)[w.o[q]] = o[q];
  }
}

// ---------------------------------------------------------------------------
// casts / re-layouts of the parameters
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
cast_f32_bf16_kernel(const float* __restrict__ src, bf16* __restrict__ dst, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = (bf16)src[i];
}

// eight elements per thread: two 16-byte loads, one 16-byte store (both bases 16-byte aligned, n8 = n / 8)
__global__ void __launch_bounds__(256)
cast_f32_bf16_x8_kernel(const float4* __restrict__ src, bf16x8* __restrict__ dst, long long n8) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n8) return;
  const float4 a = src[2 * i], b = src[2 * i + 1];
  bf16x8 o;
  o[0] = (bf16)a.x; o[1] = (bf16)a.y; o[2] = (bf16)a.z; o[3] = (bf16)a.w;
  o[4] = (bf16)b.x; o[5] = (bf16)b.y; o[6] = (bf16)b.z; o[7] = (bf16)b.w;
  dst[i] = o;
}

// dst[c][r] = bf16(src[r][c]) for a row-major [R][Ccols] float matrix (32x32 LDS tiles)
template <typename SrcT>
__global__ void __launch_bounds__(256)
transpose_cast_kernel(const SrcT* __restrict__ src, bf16* __restrict__ dst, int R, int Ccols) {
  __shared__ float tile[32][33];
  const int c0 = blockIdx.x * 32, r0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
  for (int j = ty; j < 32; j += 8)
    if (r0 + j < R && c0 + tx < Ccols) tile[j][tx] = (float)src[(long long)(r0 + j) * Ccols + c0 + tx];
  __syncthreads();
  for (int j = ty; j < 32; j += 8)
    if (c0 + j < Ccols && r0 + tx < R) dst[(long long)(c0 + j) * R + r0 + tx] = (bf16)tile[tx][j];
}

// bf16 [R][C] -> bf16 [C][R] in 64 x 64 tiles with 16-byte global accesses on both sides
// (R and C multiples of 8, both bases 16-byte aligned); the 2-byte transpose happens in LDS
__global__ void __launch_bounds__(256)
transpose_bf16_kernel(const bf16* __restrict__ src, bf16* __restrict__ dst, int R, int Ccols) {
  __shared__ bf16 tile[64][64 + 2];
  const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64;
  const int t = threadIdx.x, ch = t & 7;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int r = (t >> 3) + 32 * h;
    if (r0 + r < R && c0 + ch * 8 < Ccols) {
      const uint4 v = *reinterpret_cast<const uint4*>(src + (long long)(r0 + r) * Ccols + c0 + ch * 8);
      const bf16* e = reinterpret_cast<const bf16*>(&v);
#pragma unroll
      for (int q = 0; q < 8; ++q) tile[r][ch * 8 + q] = e[q];
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int c = (t >> 3) + 32 * h;          // output row = source column
    if (c0 + c < Ccols && r0 + ch * 8 < R) {
      uint4 v;
      bf16* e = reinterpret_cast<bf16*>(&v);
#pragma unroll
      for (int q = 0; q < 8; ++q) e[q] = tile[ch * 8 + q][c];
      *reinterpret_cast<uint4*>(dst + (long long)(c0 + c) * R + r0 + ch * 8) = v;
    }
  }
}

// conv filter HWIO [9][Cin][Cout] float -> forward operand [Cout][9*Cin] bf16 and
// data-gradient operand [Cin][9*Cout] bf16 (taps stay in forward order; the
// gradient kernel walks them backwards)
__global__ void __launch_bounds__(256)
conv_weight_layouts_kernel(const float* __restrict__ w, bf16* __restrict__ w_fwd, bf16* __restrict__ w_dgrad, int Cin,
                           int Cout) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 9LL * Cin * Cout) return;
  const int co = (int)(i % Cout);
  const int ci = (int)((i / Cout) % Cin);
  const int t = (int)(i / ((long long)Cout * Cin));
  const bf16 v = (bf16)w[i];
  if (w_fwd) w_fwd[(long long)co * 9 * Cin + t * Cin + ci] = v;
  if (w_dgrad) w_dgrad[(long long)ci * 9 * Cout + t * Cout + co] = v;
}

// The same re-layout for up to 16 layers whose filters lie in one flat parameter buffer: one launch per training
// step instead of one per layer.
struct ConvLayoutTable {
  long long src[16];      // offset of the layer's [9][Cin][Cout] filter in the flat buffer (floats)
  long long end[16];      // running total of filter elements up to and including the layer
  bf16* w_fwd[16];
  bf16* w_dgrad[16];
  int cin[16], cout[16];
  int n;
};
__global__ void __launch_bounds__(256)
conv_weight_layouts_multi_kernel(const float* __restrict__ flat, ConvLayoutTable t) {
  const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= t.end[t.n - 1]) return;
  int l = 0;
  while (g >= t.end[l]) ++l;
  const long long i = g - (l ? t.end[l - 1] : 0);
  const int Cin = t.cin[l], Cout = t.cout[l];
  const int co = (int)(i % Cout);
  const int ci = (int)((i / Cout) % Cin);
  const int tap = (int)(i / ((long long)Cout * Cin));
  const bf16 v = (bf16)flat[t.src[l] + i];
  if (t.w_fwd[l]) t.w_fwd[l][(long long)co * 9 * Cin + tap * Cin + ci] = v;
  if (t.w_dgrad[l]) t.w_dgrad[l][(long long)ci * 9 * Cout + tap * Cout + co] = v;
}

// ---------------------------------------------------------------------------
// loss (cnn_model.py:139-140): loss = 0.5 * sum(((out - gt) Wl)^2), d_out = ((out - gt) Wl) Wl^T
// ---------------------------------------------------------------------------
__global__ void l2loss_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                              const float* __restrict__ Wl, float* __restrict__ loss, float* __restrict__ d_out,
                              int Bn, int D, float scale) {
  // one block; D <= 8.  loss = scale * sum(((out - gt) Wl)^2), d_out = 2 scale ((out - gt) Wl) Wl^T
  __shared__ float part[256];
  float acc = 0.f;
  for (int b = threadIdx.x; b < Bn; b += blockDim.x) {
    float e[8], r[8];
    for (int i = 0; i < D; ++i) e[i] = out[b * D + i] - gt[b * D + i];
    for (int j = 0; j < D; ++j) {
      float s = 0.f;
      for (int i = 0; i < D; ++i) s = __builtin_fmaf(e[i], Wl[i * D + j], s);
      r[j] = s;
      acc = __builtin_fmaf(s, s, acc);
    }
    if (d_out)
      for (int i = 0; i < D; ++i) {
        float s = 0.f;
        for (int j = 0; j < D; ++j) s = __builtin_fmaf(r[j], Wl[i * D + j], s);
        d_out[b * D + i] = 2.0f * scale * s;
      }
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = scale * part[0];
}

// PoseNet loss (vgg16_posenet.py:16-25): sum((x^ - x)^2) + beta sum((q^ - q)^2), rows [x(3) q(4)]
__global__ void posenet_loss_kernel(const float* __restrict__ out, const float* __restrict__ gt,
                                    float* __restrict__ loss, float* __restrict__ d_out, int Bn, float beta) {
  __shared__ float part[256];
  float acc = 0.f;
  for (int i = threadIdx.x; i < Bn * 7; i += blockDim.x) {
    const float w = (i % 7) < 3 ? 1.0f : beta;
    const float e = out[i] - gt[i];
    acc = __builtin_fmaf(w * e, e, acc);
    if (d_out) d_out[i] = 2.0f * w * e;
  }
  part[threadIdx.x] = acc;
  __syncthreads();
  for (int o = blockDim.x / 2; o > 0; o >>= 1) {
    if (threadIdx.x < o) part[threadIdx.x] += part[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = part[0];
}

// ---------------------------------------------------------------------------
// Adam (TensorFlow AdamOptimizer defaults: beta1 0.9, beta2 0.999, eps 1e-8;
// lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)); grad_scale folds the 1/world
// of the data-parallel mean
// ---------------------------------------------------------------------------
// shadow (optional): bf16 copy of the first shadow_n updated parameters, written in the same
// pass (the fully connected weights' operand copy: saves reading them again for the cast)
__global__ void __launch_bounds__(256)
adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
            long long n, float lr_t, float beta1, float beta2, float eps, float grad_scale,
            bf16* __restrict__ shadow, long long shadow_n, const float* __restrict__ lr_t_dev) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (lr_t_dev) lr_t = *lr_t_dev;   // the step size of a launch replayed from a HIP graph lives in device memory
  const float gi = g[i] * grad_scale;
  const float mi = beta1 * m[i] + (1.0f - beta1) * gi;
  const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  const float pi = p[i] - lr_t * mi / (sqrtf(vi) + eps);
  p[i] = pi;
  if (i < shadow_n) shadow[i] = (bf16)pi;
}

// tf.train.GradientDescentOptimizer: p -= lr * g
__global__ void __launch_bounds__(256)
sgd_kernel(float* __restrict__ p, const float* __restrict__ g, long long n, float lr, float grad_scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) p[i] -= lr * (g[i] * grad_scale);
}

// tf.train.RMSPropOptimizer (momentum 0): ms = decay ms + (1 - decay) g^2; p -= lr g / sqrt(ms + eps)
__global__ void __launch_bounds__(256)
rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms, long long n, float lr,
               float decay, float eps, float grad_scale) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gi = g[i] * grad_scale;
  const float m = decay * ms[i] + (1.0f - decay) * gi * gi;
  ms[i] = m;
  p[i] -= lr * gi / sqrtf(m + eps);
}

static unsigned int blocks_for(long long n, int per = 256) { return (unsigned int)((n + per - 1) / per); }


// Grid of 256-thread blocks for a grid-stride walk over total8 8-channel vectors in which every
// thread keeps the same channels: blocks * 256 * 8 elements must be a multiple of C.
static unsigned int channel_stride_grid(long long total8, int C) {
  long long a = C, b = 2048;
  while (b) { const long long t = a % b; a = b; b = t; }      // a = gcd(C, 2048)
  const long long unit = C / a;                                // blocks per period
  long long g = (total8 + 255) / 256;
  if (g > 4096) g = 4096;
  g = g / unit * unit;
  return (unsigned int)(g < unit ? unit : g);
}

// rows are strided over the grid: enough blocks to fill the chip, at least ~8 rows per row lane
static unsigned int reduce_grid(long long P, int C) {
  const int rlanes = 256 / (C / 8);
  constexpr long long cap = 1024;
  // every block ends in 2C float atomics on the same addresses: beyond ~1024 blocks their
  // serialisation costs more than the extra parallelism brings
  long long g = (P + 8LL * rlanes - 1) / (8LL * rlanes);
  return (unsigned int)(g < 1 ? 1 : (g > cap ? cap : g));
}

// 256 x 256 tiles: wide layers with enough pixel tiles to fill most of the chip
static bool use_igemm256(long long M, int N, const void* out_bf16, const float* out_f32) {
  static const int mode = [] { const char* e = getenv("SLAMHIP_IGEMM256"); return e ? atoi(e) : 1; }();   // 0 off, 2 always
  if (mode == 0 || N % 256 != 0 || !out_bf16 || out_f32) return false;
  // one 256 x 256 workgroup per compute unit: worth it from about three quarters of the chip
  // (196 tiles of a 14 x 14 x 512 layer at batch 128: 147 us against 178 us with 128-row tiles)
  constexpr int min_tiles = 192;
  return mode == 2 || (M + 255) / 256 * (N / 256) >= min_tiles;     // 2: always (tests)
}

template <int TAPS>
static void launch_igemm256(const void* a, const void* b, void* out_bf16, long long M, int N, int C, int H, int W,
                            int flip, hipStream_t s, float* stat_part, int m_tile0 = 0, int m_tiles = -1) {
  if (m_tiles < 0) m_tiles = (int)((M + 255) / 256);
  const int n_tiles = N / 256, total = m_tiles * n_tiles;
  hipLaunchKernelGGL((igemm256_bf16_kernel<TAPS>), dim3((total + 7) / 8 * 8), dim3(512), 0, s, (const bf16*)a,
                     (const bf16*)b, (bf16*)out_bf16, (int)M, N, C, H, W, flip, n_tiles, total, stat_part, m_tile0);
}

template <int TAPS, int BN>
static void launch_igemm(const void* a, const void* b, void* out_bf16, float* out_f32, long long M, int N, int C,
                         int H, int W, int flip, hipStream_t s, float* stat_part = nullptr, int splitk = 1,
                         int m_tile0 = 0, int m_tiles = -1) {
  if (m_tiles < 0) m_tiles = (int)((M + 127) / 128);
  const int n_tiles = (N + BN - 1) / BN, total = m_tiles * n_tiles;
  const dim3 grid((total + 7) / 8 * 8, splitk), block(256);
#define SLAMHIP_IGEMM_GO(WB, WF)                                                                                 \
  hipLaunchKernelGGL((igemm_bf16_kernel<TAPS, BN, WB, WF>), grid, block, 0, s, (const bf16*)a, (const bf16*)b, \
                     (bf16*)out_bf16, out_f32, (int)M, N, C, H, W, flip, n_tiles, total, stat_part, splitk, m_tile0)
  if (out_bf16 && out_f32) SLAMHIP_IGEMM_GO(true, true);
  else if (out_bf16) SLAMHIP_IGEMM_GO(true, false);
  else SLAMHIP_IGEMM_GO(false, true);
#undef SLAMHIP_IGEMM_GO
}

// A 3x3 convolution (or its data gradient) with the tile sizes that fit it; returns the number
// of statistics records written when stat_part is given.  One 256 x 256 workgroup occupies a
// compute unit, so the 256-tile kernel runs in rounds of one tile per compute unit; when the
// last round would be mostly empty, only the full rounds go to it and the remaining pixel rows
// to the 128-tile kernel (whose records follow the first kernel's).
static long long launch_conv3x3(const void* act, const void* w_nk, void* out_bf16, float* out_f32, long long M, int N,
                                int C, int H, int W, int flip, hipStream_t s, float* stat_part) {
  if (use_igemm256(M, N, out_bf16, out_f32)) {
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
      return n;
    }();
    const int m_tiles = (int)((M + 255) / 256), n_tiles = N / 256;
    const long long total = (long long)m_tiles * n_tiles;
    const long long rem = total % cus;
    int m_big = m_tiles;
    if (total > cus && rem != 0 && rem * 4 < cus) m_big = (int)((total - rem) / n_tiles);
    launch_igemm256<9>(act, w_nk, out_bf16, M, N, C, H, W, flip, s, stat_part, 0, m_big);
    long long records = 2LL * m_big;
    if (m_big < m_tiles) {
      // the leftover pixel rows: filter-row kernel (a third of the K-steps, two in flight), which a
      // launch of a few dozen workgroups, bound by the latency of its K-steps, finishes sooner
      const int m_small = (int)((M - 256LL * m_big + 127) / 128);
      float* part = stat_part ? stat_part + records * 2 * N : nullptr;
      const int m_rest = (m_small + 1) / 2, nt = N / 64, total_rest = m_rest * nt;
      hipLaunchKernelGGL(igemm_row3w_bf16_kernel, dim3((total_rest + 7) / 8 * 8), dim3(256), 0, s, (const bf16*)act,
                         (const bf16*)w_nk, (bf16*)out_bf16, (int)M, N, C, H, W, flip, nt, total_rest, part, m_big);
      records += 2LL * m_small;
    }
    return records;
  }
  // 64 -> 64 channels with many pixels: the filter held in registers across a persistent walk over the pixel tiles,
  // one workgroup per compute unit.  Off unless SLAMHIP_CONV_WS=1: measured at 224 x 224, batch 128 it equals the
  // filter-row kernel (forward 717 against 720 us, data gradient 736 against 729 us) - with one wave per SIMD its
  // epilogue, DMA issue and fragment reads do not hide behind the MFMAs (timing ablations: MFMAs and barriers alone
  // 367 us, + epilogue 144, + DMA 137, + fragment reads 94)
  static const bool conv_ws = [] { const char* e = getenv("SLAMHIP_CONV_WS"); return e ? atoi(e) != 0 : false; }();
  if (conv_ws && out_bf16 && !out_f32 && N == 64 && C == 64 && M >= 128 * 1024 && M < (1LL << 25) && W >= 1 && H >= 256 / W + 1) {
    static const int ws_grid = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n < 8)
        n = 256;
      return n / 8 * 8;
    }();
    const int m_tiles = (int)((M + 255) / 256);
    hipLaunchKernelGGL(conv3x3_c64_ws_kernel, dim3(ws_grid), dim3(256), 0, s, (const bf16*)act, (const bf16*)w_nk,
                       (bf16*)out_bf16, (int)M, H, W, flip, m_tiles, stat_part);
    return (M + 127) / 128 * 2;
  }
  // many pixels, few channels (bfloat16 output only): one K-step per filter row (SLAMHIP_IGEMM_ROW3=0: off)
  // (measured at batch 128 against the per-tap kernel: 224 x 224 x 64 -> 64 forward 940 -> 860 us, data
  // gradient 847 -> 765 us; 112 x 112 x 128 -> 64 data gradient 434 -> 372 us; 128-wide outputs as two
  // 64-wide tiles: 405 -> 387 us at 112 x 112 x 64 -> 128, 569 -> 554 us forward and 567 -> 547 us data
  // gradient at 112 x 112 x 128 -> 128, 306 -> 275 us data gradient at 56 x 56 x 256 -> 128)
  static const bool row3 = [] { const char* e = getenv("SLAMHIP_IGEMM_ROW3"); return e ? atoi(e) != 0 : true; }();
  const bool row3_shape = N % 64 == 0 && (N % 128 != 0 || (N == 128 && (C == 64 || C == 128 || C == 256)));
  if (row3 && out_bf16 && !out_f32 && row3_shape && C <= 256 && M >= 128 * 1024) {
    const int m_tiles = (int)((M + 255) / 256), n_tiles = N / 64, total = m_tiles * n_tiles;
    hipLaunchKernelGGL(igemm_row3w_bf16_kernel, dim3((total + 7) / 8 * 8), dim3(256), 0, s, (const bf16*)act,
                       (const bf16*)w_nk, (bf16*)out_bf16, (int)M, N, C, H, W, flip, n_tiles, total, stat_part, 0);
    return (M + 127) / 128 * 2;
  }
  if (N % 128 == 0) launch_igemm<9, 128>(act, w_nk, out_bf16, out_f32, M, N, C, H, W, flip, s, stat_part);
  else launch_igemm<9, 64>(act, w_nk, out_bf16, out_f32, M, N, C, H, W, flip, s, stat_part);
  return (M + 127) / 128 * 2;
}

// Runs launch(dst, det): a kernel whose `slots` workgroups (or slices) each contribute `width` floats to `out` - with
// atomics into the cleared `out` (det = 0, dst = out), or each to its own slot of the scratch buffer, added
// afterwards in slot order (det = 1; every slot must be written).
template <typename F>
static int accumulate(float* out, long long width, long long slots, hipStream_t s, F launch) {
  if (deterministic_mode()) {
    float* part = det_scratch(sizeof(float) * (size_t)width * (size_t)slots);
    SLAMHIP_REQUIRE(part, SLAMHIP_E_HIP, "deterministic mode: no scratch memory");
    launch(part, 1);
    SLAMHIP_HIP(hipGetLastError());
    SLAMHIP_HIP(ordered_sum(part, out, slots, width, s));
    return SLAMHIP_OK;
  }
  SLAMHIP_HIP(clear_accumulator(out, sizeof(float) * (size_t)width, s));
  launch(out, 0);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

// Grid of the first layer's forward kernel, whose workgroups walk the pixel tiles: as many workgroups as the chip
// holds at once (a grid of 2048 on 1280 slots ran a second, 60 % full round)
template <int MODE>
static int c4_fwd_grid(int tiles) {
  static const int slots = [] {
    int dev = 0, cus = 0, per_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess ||
        hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
      cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv3x3_c4_k64_fwd_kernel<MODE>, 256, 0) != hipSuccess ||
        per_cu <= 0)
      per_cu = 4;
    return cus * per_cu;
  }();
  return tiles < slots ? tiles : slots;
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

int slamhip_conv3x3_igemm_bf16(const void* act, const void* w_nk, void* out_bf16, float* out_f32, int B, int H,
                               int W, int C, int N, int flip, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && N > 0, SLAMHIP_E_ARG, "conv3x3_igemm: bad shape");
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(act && w_nk && (out_bf16 || out_f32), SLAMHIP_E_ARG, "conv3x3_igemm: null pointer");
  SLAMHIP_REQUIRE(C % 64 == 0, SLAMHIP_E_SIZE, "conv3x3_igemm: channels per tap (%d) must be a multiple of 64", C);
  const long long M = (long long)B * H * W;
  SLAMHIP_REQUIRE(M <= 0x7fffffffLL, SLAMHIP_E_SIZE, "conv3x3_igemm: too many pixels");
  hipStream_t s = (hipStream_t)stream;
  launch_conv3x3(act, w_nk, out_bf16, out_f32, M, N, C, H, W, flip, s, nullptr);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

size_t slamhip_conv_stats_workspace_bytes(long long pixels, int N) {
  if (pixels <= 0 || N <= 0) return 0;
  return (size_t)((pixels + 127) / 128) * 2 * 2 * (size_t)N * sizeof(float);
}

// sums2c[0..2N) = column sums of the R per-tile records in ws
static int finish_stats(const float* ws, float* sums2c, long long R, int N, hipStream_t s) {
  long long slices = (R + 63) / 64;
  if (slices > 512) slices = 512;
  if (slices < 1) slices = 1;
  const dim3 grid((2 * N + 255) / 256, (unsigned)slices);
  if (deterministic_mode()) {      // the slices' sums to their own slots, then added in slice order
    float* part = det_scratch(sizeof(float) * 2 * (size_t)N * slices);
    SLAMHIP_REQUIRE(part, SLAMHIP_E_HIP, "deterministic mode: no scratch memory");
    hipLaunchKernelGGL(stat_finish_kernel, grid, dim3(256), 0, s, ws, part, (int)R, 2 * N, 1);
    SLAMHIP_HIP(hipGetLastError());
    SLAMHIP_HIP(ordered_sum(part, sums2c, slices, 2LL * N, s));
    return SLAMHIP_OK;
  }
  SLAMHIP_HIP(clear_accumulator(sums2c, sizeof(float) * 2 * (size_t)N, s));
  hipLaunchKernelGGL(stat_finish_kernel, grid, dim3(256), 0, s, ws, sums2c, (int)R, 2 * N, 0);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_conv3x3_igemm_bf16_stats(const void* act, const void* w_nk, void* out_bf16, int B, int H, int W, int C,
                                     int N, float* sums2c, void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && N > 0, SLAMHIP_E_ARG, "conv3x3_igemm_stats: bad shape");
  SLAMHIP_REQUIRE(act && w_nk && out_bf16 && sums2c && workspace, SLAMHIP_E_ARG, "conv3x3_igemm_stats: null pointer");
  SLAMHIP_REQUIRE(C % 64 == 0, SLAMHIP_E_SIZE, "conv3x3_igemm_stats: channels per tap (%d) must be a multiple of 64", C);
  const long long M = (long long)B * H * W;
  SLAMHIP_REQUIRE(M <= 0x7fffffffLL, SLAMHIP_E_SIZE, "conv3x3_igemm_stats: too many pixels");
  SLAMHIP_REQUIRE(workspace_bytes >= slamhip_conv_stats_workspace_bytes(M, N), SLAMHIP_E_SIZE,
                  "conv3x3_igemm_stats: workspace of %zu bytes, need %zu", workspace_bytes,
                  slamhip_conv_stats_workspace_bytes(M, N));
  hipStream_t s = (hipStream_t)stream;
  // one record per (pixel tile, wave row) of whichever kernels ran
  const long long records = launch_conv3x3(act, w_nk, out_bf16, nullptr, M, N, C, H, W, 0, s, (float*)workspace);
  SLAMHIP_HIP(hipGetLastError());
  return finish_stats((const float*)workspace, sums2c, records, N, s);
}

int slamhip_gemm_nt_bf16(const void* a_mk, const void* b_nk, void* out_bf16, float* out_f32, int M, int N, int K,
                         void* stream) {
  SLAMHIP_REQUIRE(M >= 0 && N > 0 && K > 0, SLAMHIP_E_ARG, "gemm_nt: bad shape");
  if (M == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE(a_mk && b_nk && (out_bf16 || out_f32), SLAMHIP_E_ARG, "gemm_nt: null pointer");
  SLAMHIP_REQUIRE(K % 64 == 0, SLAMHIP_E_SIZE, "gemm_nt: K (%d) must be a multiple of 64", K);
  hipStream_t s = (hipStream_t)stream;
  // few output tiles and a long K (the fully connected layers at small batch): split K over
  // blockIdx.y so that the weight stream is spread over the chip; float32 output only
  const int bn = (N % 128 == 0) ? 128 : 64;
  const long long tiles = (long long)((M + 127) / 128) * ((N + bn - 1) / bn);
  int splitk = 1;
  if (!out_bf16 && tiles < 256 && K >= 1024 && !deterministic_mode()) {   // (fixed-order mode: one slice, no atomics)
    splitk = (int)((512 + tiles - 1) / tiles);
    if (splitk > K / 256) splitk = K / 256;      // at least four K-steps per slice
    if (splitk < 1) splitk = 1;
  }
  if (splitk > 1) SLAMHIP_HIP(zero_async(out_f32, sizeof(float) * (size_t)M * N, s));
  if (bn == 128) launch_igemm<1, 128>(a_mk, b_nk, out_bf16, out_f32, M, N, K, 1, 1, 0, s, nullptr, splitk);
  else launch_igemm<1, 64>(a_mk, b_nk, out_bf16, out_f32, M, N, K, 1, 1, 0, s, nullptr, splitk);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_wgrad_bf16(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout, int taps,
                       void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && (taps == 1 || taps == 9), SLAMHIP_E_ARG,
                  "wgrad: bad shape");
  SLAMHIP_REQUIRE(x && dy && dw, SLAMHIP_E_ARG, "wgrad: null pointer");
  SLAMHIP_REQUIRE(Cin % 64 == 0 && Cout % 64 == 0, SLAMHIP_E_SIZE, "wgrad: channels must be multiples of 64");
  hipStream_t s = (hipStream_t)stream;
  const long long P = (long long)B * H * W;
  SLAMHIP_REQUIRE(P <= 0x7fffffffLL, SLAMHIP_E_SIZE, "wgrad: too many pixels");
  // wide layers: 256 x 256 tiles with DMA-staged operands, one workgroup per compute unit and tap
  // (SLAMHIP_WGRAD256 = 0 turns it off)
  static const bool wgrad256 = [] { const char* e = getenv("SLAMHIP_WGRAD256"); return e ? atoi(e) != 0 : true; }();
  if (wgrad256 && taps == 9 && Cin % 256 == 0 && Cout % 256 == 0 && P >= 64 * 64 && P * (long long)W < (1LL << 32) &&
      W >= 2 && H >= 2) {
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
        n = 256;
      return n;
    }();
    const long long jobs = 9LL * (Cin / 256) * (Cout / 256);
    long long split = cus / jobs;                       // one round of one workgroup per compute unit
    if (split > P / (64 * 8)) split = P / (64 * 8);     // at least eight K-steps per slice
    if (split < 1) split = 1;
    const long long chunk = ((P + split - 1) / split + 63) / 64 * 64;
    split = (P + chunk - 1) / chunk;                    // no empty slice
    const long long n_dw = 9LL * Cin * Cout;
    float* part = nullptr;
    if (split > 1) {
      if (deterministic_mode()) {
        part = det_scratch(sizeof(float) * (size_t)n_dw * split);
        SLAMHIP_REQUIRE(part, SLAMHIP_E_HIP, "deterministic mode: no scratch memory");
      } else {
        SLAMHIP_HIP(clear_accumulator(dw, sizeof(float) * (size_t)n_dw, s));
      }
    }
    const long long total = jobs * split;
    const unsigned int magic_w = (unsigned int)(((1ULL << 32) + (unsigned long long)W - 1) / (unsigned long long)W);
    const unsigned int magic_h = (unsigned int)(((1ULL << 32) + (unsigned long long)H - 1) / (unsigned long long)H);
    hipLaunchKernelGGL(wgrad256_bf16_kernel, dim3((unsigned)((total + 7) / 8 * 8)), dim3(512), 0, s, (const bf16*)x,
                       (const bf16*)dy, part ? part : dw, (int)P, Cin, Cout, H, W, (int)split, (int)total, magic_w,
                       magic_h, part ? n_dw : 0LL);
    SLAMHIP_HIP(hipGetLastError());
    if (part) SLAMHIP_HIP(ordered_sum(part, dw, split, n_dw, s));
    return SLAMHIP_OK;
  }
  // up to 128 input channels on wide images: one workgroup per filter row and 64 x 64 tile (see the
  // kernel; 1037 -> 726 us at 224 x 224 x 64 x 64, 772 -> 649 us at 112 x 112 x 128 x 128, batch 128).
  // SLAMHIP_WGRAD_ROWS64 = 0 turns it off, n limits it to 64 n input channels.
  static const int rows64 = [] { const char* e = getenv("SLAMHIP_WGRAD_ROWS64"); return e ? atoi(e) : 2; }();
  if (rows64 && taps == 9 && Cin % 64 == 0 && Cin <= 64 * rows64 && W % 112 == 0 && P > 0) {
    const long long tiles3 = 3LL * (Cin / 64) * (Cout / 64);
    const long long S = (long long)B * H * (W / 112);
    long long split = 1536 / tiles3;                  // two rounds at three workgroups per compute unit
    if (split > S / 16) split = S / 16;               // at least 16 segments per slice
    if (split < 1) split = 1;
    const long long chunk = (S + split - 1) / split;
    split = (S + chunk - 1) / chunk;
    const long long n_dw = 9LL * Cin * Cout;
    float* part = nullptr;
    if (deterministic_mode()) {
      part = det_scratch(sizeof(float) * (size_t)n_dw * split);
      SLAMHIP_REQUIRE(part, SLAMHIP_E_HIP, "deterministic mode: no scratch memory");
    } else {
      SLAMHIP_HIP(clear_accumulator(dw, sizeof(float) * (size_t)n_dw, s));
    }
    const long long total3 = tiles3 * split;
    hipLaunchKernelGGL(wgrad_rows64_kernel, dim3((unsigned)((total3 + 7) / 8 * 8)), dim3(256), 0, s, (const bf16*)x,
                       (const bf16*)dy, part ? part : dw, B, H, W, Cin, Cout, (int)split, (int)total3, part ? n_dw : 0LL);
    SLAMHIP_HIP(hipGetLastError());
    if (part) SLAMHIP_HIP(ordered_sum(part, dw, split, n_dw, s));
    return SLAMHIP_OK;
  }
  // narrow images (56, 28, 14 wide here): groups of whole image rows per K-step, see the kernel.
  // SLAMHIP_WGRAD_ROWGROUP = 0 turns it off, otherwise it is the number of workgroups aimed at.
  static const int rowgroup = [] { const char* e = getenv("SLAMHIP_WGRAD_ROWGROUP"); return e ? atoi(e) : 512; }();
  if (rowgroup && taps == 9 && W + 2 <= 128 && P > 0) {
    const int G = 128 / (W + 2);
    const long long BH = (long long)B * H;
    if (G * W >= 96 && BH >= G) {                          // at least three quarters of a K-step carry pixels
      const int TMg = Cin % 128 == 0 ? 128 : 64;
      const long long tiles3 = 3LL * (Cin / TMg) * (Cout / 64);
      const long long S = (BH + G - 1) / G;
      long long split = rowgroup / tiles3;
      if (split > S / 8) split = S / 8;                 // at least 8 K-steps per slice
      if (split < 1) split = 1;
      const long long chunk = (S + split - 1) / split;
      split = (S + chunk - 1) / chunk;
      const long long n_dw = 9LL * Cin * Cout;
      float* part = nullptr;
      if (deterministic_mode()) {
        part = det_scratch(sizeof(float) * (size_t)n_dw * split);
        SLAMHIP_REQUIRE(part, SLAMHIP_E_HIP, "deterministic mode: no scratch memory");
      } else {
        SLAMHIP_HIP(clear_accumulator(dw, sizeof(float) * (size_t)n_dw, s));
      }
      float* const dst = part ? part : dw;
      const long long stride = part ? n_dw : 0LL;
      const long long total3 = tiles3 * split;
      const dim3 grid3((unsigned)((total3 + 7) / 8 * 8));
      if (TMg == 128)
        hipLaunchKernelGGL(wgrad_rowgroup_kernel<128>, grid3, dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, dst,
                           (int)BH, H, W, Cin, Cout, G, (int)split, (int)total3, stride);
      else
        hipLaunchKernelGGL(wgrad_rowgroup_kernel<64>, grid3, dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, dst,
                           (int)BH, H, W, Cin, Cout, G, (int)split, (int)total3, stride);
      SLAMHIP_HIP(hipGetLastError());
      if (part) SLAMHIP_HIP(ordered_sum(part, dw, split, n_dw, s));
      return SLAMHIP_OK;
    }
  }
  const int TM = Cin % 128 == 0 ? 128 : 64, TN = Cout % 128 == 0 ? 128 : 64;
  const long long tiles = (long long)(Cin / TM) * (Cout / TN) * taps;
  constexpr long long target_blocks = 2048;
  // at most target_blocks workgroups (two full rounds at four per compute unit: a few more would
  // start a third, nearly empty round), at least 3584 pixels per slice (each slice ends in
  // TM x TN float atomics)
  long long splitk = target_blocks / tiles;
  const long long max_split = (P + 3583) / 3584;
  if (splitk > max_split) splitk = max_split;
  if (splitk < 1) splitk = 1;
  // slices are rounded up to 64 pixels; drop the ones that rounding leaves empty
  if (P > 0) {
    const long long chunk = ((P + splitk - 1) / splitk + 63) / 64 * 64;
    splitk = (P + chunk - 1) / chunk;
  }
  const long long n_dw = (long long)taps * Cin * Cout;
  if (P == 0) {
    SLAMHIP_HIP(zero_async(dw, sizeof(float) * (size_t)n_dw, s));
    return SLAMHIP_OK;
  }
  float* part = nullptr;
  if (splitk != 1) {
    if (deterministic_mode()) {
      part = det_scratch(sizeof(float) * (size_t)n_dw * splitk);
      SLAMHIP_REQUIRE(part, SLAMHIP_E_HIP, "deterministic mode: no scratch memory");
    } else {
      SLAMHIP_HIP(clear_accumulator(dw, sizeof(float) * (size_t)n_dw, s));
    }
  }
  float* const dst = part ? part : dw;
  const long long stride = part ? n_dw : 0LL;
  const long long total = tiles * splitk;
  SLAMHIP_REQUIRE(total < (1LL << 30), SLAMHIP_E_SIZE, "wgrad: too many workgroups");
  const dim3 grid((unsigned)((total + 7) / 8 * 8));
#define SLAMHIP_WGRAD_GO(TAPS_, TM_, TN_)                                                                          \
  hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS_, TM_, TN_>), grid, dim3(256), 0, s, (const bf16*)x, (const bf16*)dy, \
                     dst, (int)P, Cin, Cout, H, W, (int)splitk, (int)total, stride)
  if (taps == 9) {
    if (TM == 128 && TN == 128) SLAMHIP_WGRAD_GO(9, 128, 128);
    else if (TM == 64 && TN == 128) SLAMHIP_WGRAD_GO(9, 64, 128);
    else if (TM == 128 && TN == 64) SLAMHIP_WGRAD_GO(9, 128, 64);
    else SLAMHIP_WGRAD_GO(9, 64, 64);
  } else {
    if (TM == 128 && TN == 128) SLAMHIP_WGRAD_GO(1, 128, 128);
    else if (TM == 64 && TN == 128) SLAMHIP_WGRAD_GO(1, 64, 128);
    else if (TM == 128 && TN == 64) SLAMHIP_WGRAD_GO(1, 128, 64);
    else SLAMHIP_WGRAD_GO(1, 64, 64);
  }
#undef SLAMHIP_WGRAD_GO
  SLAMHIP_HIP(hipGetLastError());
  if (part) SLAMHIP_HIP(ordered_sum(part, dw, splitk, n_dw, s));
  return SLAMHIP_OK;
}

static int small_fwd(const void* x, const float* w_hwio, void* out_bf16, int B, int H, int W, int Cin, int Cout,
                     float* stat_part, hipStream_t s) {
  const long long total = (long long)B * H * W * Cout;
  const int P = B * H * W;
  if (Cin == 4 && Cout == 64) {
    const int tiles = (P + 127) / 128;
    hipLaunchKernelGGL(conv3x3_c4_k64_fwd_kernel<0>, dim3(c4_fwd_grid<0>(tiles)), dim3(256), 0, s, (const bf16*)x,
                       w_hwio, (bf16*)out_bf16, P, H, W, stat_part, tiles, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, 0.0f, 0.0f);
  } else {
    hipLaunchKernelGGL(conv3x3_small_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, s, (const bf16*)x, w_hwio,
                       (bf16*)out_bf16, P, H, W, Cin, Cout);
  }
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_conv3x3_small_fwd(const void* x, const float* w_hwio, void* out_bf16, int B, int H, int W, int Cin,
                              int Cout, void* stream) {
  SLAMHIP_REQUIRE(x && w_hwio && out_bf16 && B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, SLAMHIP_E_ARG,
                  "conv3x3_small_fwd: bad argument");
  if (B == 0) return SLAMHIP_OK;
  return small_fwd(x, w_hwio, out_bf16, B, H, W, Cin, Cout, nullptr, (hipStream_t)stream);
}

int slamhip_conv3x3_small_fwd_stats(const void* x, const float* w_hwio, void* out_bf16, int B, int H, int W, int Cin,
                                    int Cout, float* sums2c, void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(x && w_hwio && out_bf16 && sums2c && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 &&
                      Cout % 64 == 0 && Cout <= 2048,
                  SLAMHIP_E_ARG, "conv3x3_small_fwd_stats: bad argument");
  hipStream_t s = (hipStream_t)stream;
  const long long P = (long long)B * H * W;
  if (Cin == 4 && Cout == 64) {
    SLAMHIP_REQUIRE(workspace && workspace_bytes >= slamhip_conv_stats_workspace_bytes(P, Cout), SLAMHIP_E_SIZE,
                    "conv3x3_small_fwd_stats: workspace of %zu bytes, need %zu", workspace_bytes,
                    slamhip_conv_stats_workspace_bytes(P, Cout));
    const int rc = small_fwd(x, w_hwio, out_bf16, B, H, W, Cin, Cout, (float*)workspace, s);
    if (rc != SLAMHIP_OK) return rc;
    return finish_stats((const float*)workspace, sums2c, (P + 127) / 128 * 2, Cout, s);
  }
  // other channel counts: the direct kernel, then a pass over its output
  const int rc = small_fwd(x, w_hwio, out_bf16, B, H, W, Cin, Cout, nullptr, s);
  if (rc != SLAMHIP_OK) return rc;
  return slamhip_channel_stats(out_bf16, sums2c, P, Cout, stream);
}

int slamhip_conv3x3_small_wgrad(const void* x, const void* dy, float* dw, int B, int H, int W, int Cin, int Cout,
                                void* stream) {
  SLAMHIP_REQUIRE(x && dy && dw && B >= 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && Cout <= 256 && 256 % Cout == 0,
                  SLAMHIP_E_ARG, "conv3x3_small_wgrad: bad argument (Cout must divide 256)");
  hipStream_t s = (hipStream_t)stream;
  const long long n_dw = 9LL * Cin * Cout;
  if (B == 0) {
    SLAMHIP_HIP(zero_async(dw, sizeof(float) * (size_t)n_dw, s));
    return SLAMHIP_OK;
  }
  const int P = B * H * W;
  if (Cin == 4 && Cout == 64) {
    // about 2048 workgroups, slices in multiples of 64 pixels
    const int px_per_block = ((P + 2047) / 2048 + 63) / 64 * 64;
    const int blocks = (P + px_per_block - 1) / px_per_block;
    return accumulate(dw, n_dw, blocks, s, [&](float* dst, int det) {
      hipLaunchKernelGGL(conv3x3_c4_k64_wgrad_kernel<false>, dim3(blocks), dim3(256), 0, s, (const bf16*)x,
                         (const bf16*)dy, dst, P, H, W, px_per_block, (const bf16*)nullptr, (const float*)nullptr,
                         (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, 0.0f, 0.0f,
                         det ? n_dw : 0LL);
    });
  }
  return accumulate(dw, n_dw, 64LL * (256 / Cout), s, [&](float* dst, int det) {
    hipLaunchKernelGGL(conv3x3_small_wgrad_kernel, dim3(9 * Cin, 64), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy,
                       dst, P, H, W, Cin, Cout, det ? n_dw : 0LL);
  });
}

int slamhip_channel_stats(const void* z, float* sums2c, long long P, int C, void* stream) {
  SLAMHIP_REQUIRE(z && sums2c && P >= 0 && C > 0 && C % 64 == 0, SLAMHIP_E_ARG, "channel_stats: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {
    SLAMHIP_HIP(zero_async(sums2c, sizeof(float) * 2 * (size_t)C, s));
    return SLAMHIP_OK;
  }
  SLAMHIP_REQUIRE(P <= 0x7fffffffLL, SLAMHIP_E_SIZE, "channel_stats: too many rows");
  SLAMHIP_REQUIRE(C <= 2048, SLAMHIP_E_SIZE, "channel_stats: more than 2048 channels");
  const unsigned grid = reduce_grid(P, C);
  return accumulate(sums2c, 2LL * C, grid, s, [&](float* dst, int det) {
    hipLaunchKernelGGL(channel_reduce_kernel<0>, dim3(grid), dim3(256), 0, s, (const bf16*)z, (const bf16*)nullptr,
                       (const bf16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dst,
                       (int)P, C, 0.0f, 0.0f, det);
  });
}

int slamhip_bn_relu_fwd(const void* z, const float* sums2c, const float* gamma, const float* beta, void* y,
                        long long P, int C, float eps, void* stream) {
  SLAMHIP_REQUIRE(z && sums2c && gamma && beta && y && P > 0 && C > 0 && C % 8 == 0, SLAMHIP_E_ARG,
                  "bn_relu_fwd: bad argument");
  const long long total8 = P * C / 8;
  hipLaunchKernelGGL(bn_relu_fwd_kernel, dim3(channel_stride_grid(total8, C)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)z, sums2c, gamma, beta, (bf16*)y, total8, C, 1.0f / (float)P, eps);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

// phases: bit 0 = the reduction pass (red2c), bit 1 = the pass that writes dz from red2c.  A caller that keeps
// the normalisation statistics over several ranks runs the two passes apart and averages red2c over the
// ranks in between (sums2c averaged the same way after the forward's statistics).
int slamhip_bn_relu_bwd_phases(const void* dy, const void* z, const float* sums2c, const float* gamma,
                               const float* beta, float* red2c, void* dz, long long P, int C, float eps, int phases,
                               void* stream) {
  SLAMHIP_REQUIRE(dy && z && sums2c && gamma && beta && red2c && (dz || !(phases & 2)) && P > 0 && C > 0 &&
                      C % 64 == 0 && (phases & 3) && !(phases & ~3),
                  SLAMHIP_E_ARG, "bn_relu_bwd: bad argument");
  SLAMHIP_REQUIRE(P <= 0x7fffffffLL, SLAMHIP_E_SIZE, "bn_relu_bwd: too many rows");
  SLAMHIP_REQUIRE(C <= 2048, SLAMHIP_E_SIZE, "bn_relu_bwd: more than 2048 channels");
  hipStream_t s = (hipStream_t)stream;
  if (phases & 1) {
    const unsigned grid = reduce_grid(P, C);
    const int rc = accumulate(red2c, 2LL * C, grid, s, [&](float* dst, int det) {
      hipLaunchKernelGGL(channel_reduce_kernel<1>, dim3(grid), dim3(256), 0, s, (const bf16*)dy, (const bf16*)nullptr,
                         (const bf16*)z, sums2c, gamma, beta, dst, (int)P, C, 1.0f / (float)P, eps, det);
    });
    if (rc != SLAMHIP_OK) return rc;
  }
  if (phases & 2) {
    const long long total8 = P * C / 8;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel, dim3(channel_stride_grid(total8, C)), dim3(256), 0, s, (const bf16*)dy,
                       (const bf16*)z, sums2c, red2c, gamma, beta, (bf16*)dz, total8, C, 1.0f / (float)P, eps);
  }
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bn_relu_bwd(const void* dy, const void* z, const float* sums2c, const float* gamma, const float* beta,
                        float* red2c, void* dz, long long P, int C, float eps, void* stream) {
  return slamhip_bn_relu_bwd_phases(dy, z, sums2c, gamma, beta, red2c, dz, P, C, eps, 3, stream);
}

int slamhip_bn_relu_bwd_first_layer(const void* dy, const void* z, const float* sums2c, const float* gamma,
                                    const float* beta, float* red2c, const void* x, float* dw, int B, int H, int W,
                                    float eps, void* stream) {
  SLAMHIP_REQUIRE(dy && z && sums2c && gamma && beta && red2c && x && dw && B > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "bn_relu_bwd_first_layer: bad argument");
  const long long Pl = (long long)B * H * W;
  SLAMHIP_REQUIRE(Pl <= 0x7fffffffLL / 64, SLAMHIP_E_SIZE, "bn_relu_bwd_first_layer: too many pixels");
  const int P = (int)Pl, C = 64;
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = reduce_grid(P, C);
  const int rc = accumulate(red2c, 2LL * C, grid, s, [&](float* dst, int det) {
    hipLaunchKernelGGL(channel_reduce_kernel<1>, dim3(grid), dim3(256), 0, s, (const bf16*)dy, (const bf16*)nullptr,
                       (const bf16*)z, sums2c, gamma, beta, dst, P, C, 1.0f / (float)P, eps, det);
  });
  if (rc != SLAMHIP_OK) return rc;
  const int px_per_block = ((P + 2047) / 2048 + 63) / 64 * 64;
  const int blocks = (P + px_per_block - 1) / px_per_block;
  const long long n_dw = 9LL * 4 * C;
  return accumulate(dw, n_dw, blocks, s, [&](float* dst, int det) {
    hipLaunchKernelGGL(conv3x3_c4_k64_wgrad_kernel<true>, dim3(blocks), dim3(256), 0, s, (const bf16*)x,
                       (const bf16*)dy, dst, P, H, W, px_per_block, (const bf16*)z, sums2c, (const float*)red2c, gamma,
                       beta, 1.0f / (float)P, eps, det ? n_dw : 0LL);
  });
}

int slamhip_conv3x3_c4_bn_relu_fwd(const void* x, const float* w_hwio, const float* gamma, const float* beta,
                                   void* y_bf16, float* sums2c, void* workspace, size_t workspace_bytes, int B, int H,
                                   int W, float eps, void* stream) {
  SLAMHIP_REQUIRE(x && w_hwio && gamma && beta && y_bf16 && sums2c && workspace && B > 0 && H > 0 && W > 0,
                  SLAMHIP_E_ARG, "conv3x3_c4_bn_relu_fwd: bad argument");
  const long long Pl = (long long)B * H * W;
  SLAMHIP_REQUIRE(Pl <= 0x7fffffffLL / 64, SLAMHIP_E_SIZE, "conv3x3_c4_bn_relu_fwd: too many pixels");
  SLAMHIP_REQUIRE(workspace_bytes >= slamhip_conv_stats_workspace_bytes(Pl, 64), SLAMHIP_E_SIZE,
                  "conv3x3_c4_bn_relu_fwd: workspace of %zu bytes, need %zu", workspace_bytes,
                  slamhip_conv_stats_workspace_bytes(Pl, 64));
  const int P = (int)Pl, tiles = (P + 127) / 128;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_c4_k64_fwd_kernel<1>, dim3(c4_fwd_grid<1>(tiles)), dim3(256), 0, s, (const bf16*)x, w_hwio, (bf16*)nullptr, P, H,
                     W, (float*)workspace, tiles, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr,
                     0.0f, 0.0f);
  SLAMHIP_HIP(hipGetLastError());
  const int rc = finish_stats((const float*)workspace, sums2c, (Pl + 127) / 128 * 2, 64, s);
  if (rc != SLAMHIP_OK) return rc;
  hipLaunchKernelGGL(conv3x3_c4_k64_fwd_kernel<2>, dim3(c4_fwd_grid<2>(tiles)), dim3(256), 0, s, (const bf16*)x, w_hwio, (bf16*)y_bf16, P, H,
                     W, (float*)nullptr, tiles, (const float*)sums2c, gamma, beta, 1.0f / (float)P, eps);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bn_relu_bwd_first_layer_recompute(const void* dy, const void* x, const float* w_hwio,
                                              const float* sums2c, const float* gamma, const float* beta,
                                              float* red2c, float* dw, void* workspace, size_t workspace_bytes,
                                              int B, int H, int W, float eps, void* stream) {
  SLAMHIP_REQUIRE(dy && x && w_hwio && sums2c && gamma && beta && red2c && dw && workspace && B > 0 && H > 0 && W > 0,
                  SLAMHIP_E_ARG, "bn_relu_bwd_first_layer_recompute: bad argument");
  const long long Pl = (long long)B * H * W;
  SLAMHIP_REQUIRE(Pl <= 0x7fffffffLL / 64, SLAMHIP_E_SIZE, "bn_relu_bwd_first_layer_recompute: too many pixels");
  const int P = (int)Pl, C = 64;
  const int px_per_block = ((P + 2047) / 2048 + 63) / 64 * 64;
  const int blocks = (P + px_per_block - 1) / px_per_block;
  SLAMHIP_REQUIRE(workspace_bytes >= sizeof(float) * 128 * (size_t)blocks, SLAMHIP_E_SIZE,
                  "bn_relu_bwd_first_layer_recompute: workspace of %zu bytes, need %zu", workspace_bytes,
                  sizeof(float) * 128 * (size_t)blocks);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_c4_k64_bwd_kernel<0>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy,
                     w_hwio, (float*)nullptr, (float*)workspace, P, H, W, px_per_block, sums2c, (const float*)nullptr,
                     gamma, beta, 1.0f / (float)P, eps, 0LL);
  SLAMHIP_HIP(hipGetLastError());
  const int rc = finish_stats((const float*)workspace, red2c, blocks, C, s);
  if (rc != SLAMHIP_OK) return rc;
  const long long n_dw = 9LL * 4 * C;
  return accumulate(dw, n_dw, blocks, s, [&](float* dst, int det) {
    hipLaunchKernelGGL(conv3x3_c4_k64_bwd_kernel<1>, dim3(blocks), dim3(256), 0, s, (const bf16*)x, (const bf16*)dy,
                       w_hwio, dst, (float*)nullptr, P, H, W, px_per_block, sums2c, (const float*)red2c, gamma, beta,
                       1.0f / (float)P, eps, det ? n_dw : 0LL);
  });
}

int slamhip_bn_relu_pool_fwd(const void* z, const float* sums2c, const float* gamma, const float* beta, void* y,
                             void* pooled, int B, int H, int W, int C, float eps, void* stream) {
  SLAMHIP_REQUIRE(z && sums2c && gamma && beta && pooled && B > 0 && H > 0 && W > 0 && H % 2 == 0 && W % 2 == 0 &&
                      C > 0 && C % 8 == 0,
                  SLAMHIP_E_ARG, "bn_relu_pool_fwd: bad argument (even height and width, channels % 8 == 0)");
  const long long P = (long long)B * H * W, total8 = P / 4 * (C / 8);
  hipLaunchKernelGGL(bn_relu_pool_fwd_kernel, dim3(channel_stride_grid(total8, C)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)z, sums2c, gamma, beta, (bf16*)y, (bf16*)pooled, total8, H, W, C, 1.0f / (float)P, eps);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bn_relu_pool_bwd_phases(const void* dpooled, const void* z, const float* sums2c, const float* gamma,
                                    const float* beta, float* red2c, void* dz, int B, int H, int W, int C, float eps,
                                    int phases, void* stream) {
  SLAMHIP_REQUIRE(dpooled && z && sums2c && gamma && beta && red2c && (dz || !(phases & 2)) && B > 0 && H > 0 &&
                      W > 0 && H % 2 == 0 && W % 2 == 0 && C > 0 && C % 64 == 0 && C <= 2048 && (phases & 3) &&
                      !(phases & ~3),
                  SLAMHIP_E_ARG, "bn_relu_pool_bwd: bad argument (even height and width, channels % 64 == 0, <= 2048)");
  const long long P = (long long)B * H * W, PP = P / 4;
  SLAMHIP_REQUIRE(P <= 0x7fffffffLL, SLAMHIP_E_SIZE, "bn_relu_pool_bwd: too many pixels");
  hipStream_t s = (hipStream_t)stream;
  if (phases & 1) {
    const unsigned grid = reduce_grid(PP, C);
    const int rc = accumulate(red2c, 2LL * C, grid, s, [&](float* dst, int det) {
      hipLaunchKernelGGL(bn_pool_bwd_reduce_kernel, dim3(grid), dim3(256), 0, s, (const bf16*)dpooled, (const bf16*)z,
                         sums2c, gamma, beta, dst, PP, H, W, C, 1.0f / (float)P, eps, det);
    });
    if (rc != SLAMHIP_OK) return rc;
  }
  if (phases & 2) {
    const long long total8 = PP * (C / 8);
    hipLaunchKernelGGL(bn_pool_bwd_apply_kernel, dim3(channel_stride_grid(total8, C)), dim3(256), 0, s,
                       (const bf16*)dpooled, (const bf16*)z, sums2c, red2c, gamma, beta, (bf16*)dz, total8, H, W, C,
                       1.0f / (float)P, eps);
  }
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bn_relu_pool_bwd(const void* dpooled, const void* z, const float* sums2c, const float* gamma,
                             const float* beta, float* red2c, void* dz, int B, int H, int W, int C, float eps,
                             void* stream) {
  return slamhip_bn_relu_pool_bwd_phases(dpooled, z, sums2c, gamma, beta, red2c, dz, B, H, W, C, eps, 3, stream);
}

int slamhip_bias_act_fwd(const float* z, const float* bias, void* y_bf16, float* y_f32, int M, int N, int relu,
                         void* stream) {
  SLAMHIP_REQUIRE(z && bias && (y_bf16 || y_f32) && M >= 0 && N > 0, SLAMHIP_E_ARG, "bias_act_fwd: bad argument");
  const long long total = (long long)M * N;
  if (total == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(bias_act_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, z, bias,
                     (bf16*)y_bf16, y_f32, total, N, relu);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bias_act_bwd(const float* dy, const void* y_bf16, void* dz_bf16, float* dbias, int M, int N, int relu,
                         void* stream) {
  SLAMHIP_REQUIRE(dy && dz_bf16 && dbias && (y_bf16 || !relu) && M >= 0 && N > 0, SLAMHIP_E_ARG,
                  "bias_act_bwd: bad argument");
  hipLaunchKernelGGL(bias_act_bwd_kernel, dim3(blocks_for(N, 64)), dim3(256), 0, (hipStream_t)stream, dy,
                     (const bf16*)y_bf16, (bf16*)dz_bf16, dbias, M, N, relu);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream) {
  SLAMHIP_REQUIRE(x && y && B >= 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, SLAMHIP_E_ARG, "maxpool2_fwd: bad argument");
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / 8);
  if (total == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(maxpool2_fwd_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16*)x,
                     (bf16*)y, B, H, W, C);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_maxpool2_bwd(const void* x, const void* y, const void* dy, void* dx, int B, int H, int W, int C,
                         void* stream) {
  SLAMHIP_REQUIRE(x && y && dy && dx && B >= 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0, SLAMHIP_E_ARG,
                  "maxpool2_bwd: bad argument");
  hipStream_t s = (hipStream_t)stream;
  if ((H & 1) || (W & 1)) SLAMHIP_HIP(zero_async(dx, 2 * (size_t)B * H * W * C, s));
  const long long total = (long long)B * (H / 2) * (W / 2) * (C / 8);
  if (total == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(maxpool2_bwd_kernel, dim3(blocks_for(total)), dim3(256), 0, s, (const bf16*)x, (const bf16*)y,
                     (const bf16*)dy, (bf16*)dx, B, H, W, C);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_cast_f32_bf16(const float* src, void* dst, long long n, void* stream) {
  SLAMHIP_REQUIRE(src && dst && n >= 0, SLAMHIP_E_ARG, "cast_f32_bf16: bad argument");
  if (n == 0) return SLAMHIP_OK;
  long long n8 = 0;
  if (reinterpret_cast<uintptr_t>(src) % 16 == 0 && reinterpret_cast<uintptr_t>(dst) % 16 == 0) n8 = n / 8;
  if (n8)
    hipLaunchKernelGGL(cast_f32_bf16_x8_kernel, dim3(blocks_for(n8)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)src, (bf16x8*)dst, n8);
  if (n - 8 * n8)
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3(blocks_for(n - 8 * n8)), dim3(256), 0, (hipStream_t)stream,
                       src + 8 * n8, (bf16*)dst + 8 * n8, n - 8 * n8);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_transpose_cast_bf16(const float* src_rc, void* dst_cr, int R, int C, void* stream) {
  SLAMHIP_REQUIRE(src_rc && dst_cr && R > 0 && C > 0, SLAMHIP_E_ARG, "transpose_cast: bad argument");
  hipLaunchKernelGGL(transpose_cast_kernel<float>, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0,
                     (hipStream_t)stream, src_rc, (bf16*)dst_cr, R, C);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_transpose_bf16(const void* src_rc, void* dst_cr, int R, int C, void* stream) {
  SLAMHIP_REQUIRE(src_rc && dst_cr && R > 0 && C > 0, SLAMHIP_E_ARG, "transpose_bf16: bad argument");
  const bool vec = R % 8 == 0 && C % 8 == 0 && reinterpret_cast<uintptr_t>(src_rc) % 16 == 0 &&
                   reinterpret_cast<uintptr_t>(dst_cr) % 16 == 0;
  if (vec)
    hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (R + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                       (const bf16*)src_rc, (bf16*)dst_cr, R, C);
  else
    hipLaunchKernelGGL(transpose_cast_kernel<bf16>, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0,
                       (hipStream_t)stream, (const bf16*)src_rc, (bf16*)dst_cr, R, C);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_conv_weight_layouts(const float* w_hwio, void* w_fwd, void* w_dgrad, int Cin, int Cout, void* stream) {
  SLAMHIP_REQUIRE(w_hwio && (w_fwd || w_dgrad) && Cin > 0 && Cout > 0, SLAMHIP_E_ARG, "conv_weight_layouts: bad argument");
  hipLaunchKernelGGL(conv_weight_layouts_kernel, dim3(blocks_for(9LL * Cin * Cout)), dim3(256), 0,
                     (hipStream_t)stream, w_hwio, (bf16*)w_fwd, (bf16*)w_dgrad, Cin, Cout);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_conv_weight_layouts_multi(const float* flat_params, int n_layers, const long long* offsets,
                                      const int* cin, const int* cout, void* const* w_fwd, void* const* w_dgrad,
                                      void* stream) {
  SLAMHIP_REQUIRE(flat_params && offsets && cin && cout && w_fwd && w_dgrad && n_layers >= 0 && n_layers <= 16,
                  SLAMHIP_E_ARG, "conv_weight_layouts_multi: bad argument (at most 16 layers)");
  if (n_layers == 0) return SLAMHIP_OK;
  ConvLayoutTable t;
  long long total = 0;
  for (int l = 0; l < 16; ++l) {
    const bool live = l < n_layers;
    SLAMHIP_REQUIRE(!live || (offsets[l] >= 0 && cin[l] > 0 && cout[l] > 0 && (w_fwd[l] || w_dgrad[l])), SLAMHIP_E_ARG,
                    "conv_weight_layouts_multi: bad layer %d", l);
    if (live) total += 9LL * cin[l] * cout[l];
    t.src[l] = live ? offsets[l] : 0;
    t.end[l] = total;
    t.w_fwd[l] = live ? (bf16*)w_fwd[l] : nullptr;
    t.w_dgrad[l] = live ? (bf16*)w_dgrad[l] : nullptr;
    t.cin[l] = live ? cin[l] : 1;
    t.cout[l] = live ? cout[l] : 1;
  }
  t.n = n_layers;
  hipLaunchKernelGGL(conv_weight_layouts_multi_kernel, dim3(blocks_for(total)), dim3(256), 0, (hipStream_t)stream,
                     flat_params, t);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_weighted_sq_loss(const float* out, const float* gt, const float* loss_weight, float scale, float* loss,
                             float* d_out, int B, int D, void* stream) {
  SLAMHIP_REQUIRE(out && gt && loss_weight && loss && B >= 0 && D > 0 && D <= 8, SLAMHIP_E_ARG,
                  "weighted_sq_loss: bad argument (output width must be <= 8)");
  hipLaunchKernelGGL(l2loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, gt, loss_weight, loss, d_out, B, D,
                     scale);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_l2loss(const float* out, const float* gt, const float* loss_weight, float* loss, float* d_out, int B, int D,
                   void* stream) {
  return slamhip_weighted_sq_loss(out, gt, loss_weight, 0.5f, loss, d_out, B, D, stream);
}

int slamhip_posenet_loss(const float* out7, const float* gt7, float beta, float* loss, float* d_out, int B,
                         void* stream) {
  SLAMHIP_REQUIRE(out7 && gt7 && loss && B >= 0, SLAMHIP_E_ARG, "posenet_loss: bad argument");
  hipLaunchKernelGGL(posenet_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out7, gt7, loss, d_out, B, beta);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bias_relu_fwd(const void* z, const float* bias, void* y, long long P, int C, void* stream) {
  SLAMHIP_REQUIRE(z && bias && y && P >= 0 && C > 0 && C % 8 == 0, SLAMHIP_E_ARG, "bias_relu_fwd: bad argument");
  const long long total8 = P * C / 8;
  if (total8 == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(bias_relu_nhwc_fwd_kernel, dim3(blocks_for(total8)), dim3(256), 0, (hipStream_t)stream,
                     (const bf16*)z, bias, (bf16*)y, total8, C);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_bias_relu_bwd(const void* dy, const void* y, void* dz, float* dbias, long long P, int C, void* stream) {
  SLAMHIP_REQUIRE(dy && y && dz && dbias && P >= 0 && C > 0 && C % 64 == 0 && C <= 2048, SLAMHIP_E_ARG,
                  "bias_relu_bwd: bad argument");
  SLAMHIP_REQUIRE(P <= 0x7fffffffLL, SLAMHIP_E_SIZE, "bias_relu_bwd: too many rows");
  hipStream_t s = (hipStream_t)stream;
  if (P == 0) {
    SLAMHIP_HIP(zero_async(dbias, sizeof(float) * (size_t)C, s));
    return SLAMHIP_OK;
  }
  const unsigned grid = reduce_grid(P, C);
  const int rc = accumulate(dbias, C, grid, s, [&](float* dst, int det) {
    hipLaunchKernelGGL(channel_reduce_kernel<2>, dim3(grid), dim3(256), 0, s, (const bf16*)dy, (const bf16*)y,
                       (const bf16*)nullptr, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dst,
                       (int)P, C, 0.0f, 0.0f, det);
  });
  if (rc != SLAMHIP_OK) return rc;
  const long long total8 = P * C / 8;
  hipLaunchKernelGGL(relu_gate_kernel, dim3(blocks_for(total8)), dim3(256), 0, s, (const bf16*)dy, (const bf16*)y,
                     (bf16*)dz, total8);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_adam_step(float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1,
                      float beta2, float eps, int step, float grad_scale, void* stream) {
  SLAMHIP_REQUIRE(param && grad && m && v && n >= 0 && step >= 1, SLAMHIP_E_ARG, "adam_step: bad argument");
  if (n == 0) return SLAMHIP_OK;
  const float lr_t = lr * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr_t,
                     beta1, beta2, eps, grad_scale, (bf16*)nullptr, 0LL, (const float*)nullptr);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_adam_step_shadow(float* param, const float* grad, float* m, float* v, long long n, float lr, float beta1,
                             float beta2, float eps, int step, float grad_scale, void* shadow_bf16,
                             long long shadow_n, void* stream) {
  SLAMHIP_REQUIRE(param && grad && m && v && n >= 0 && step >= 1, SLAMHIP_E_ARG, "adam_step_shadow: bad argument");
  SLAMHIP_REQUIRE(shadow_n >= 0 && shadow_n <= n && (shadow_bf16 || shadow_n == 0), SLAMHIP_E_ARG,
                  "adam_step_shadow: shadow of %lld elements for %lld parameters", shadow_n, n);
  if (n == 0) return SLAMHIP_OK;
  const float lr_t = lr * sqrtf(1.0f - powf(beta2, (float)step)) / (1.0f - powf(beta1, (float)step));
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, lr_t,
                     beta1, beta2, eps, grad_scale, (bf16*)shadow_bf16, shadow_n, (const float*)nullptr);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_adam_step_dev(float* param, const float* grad, float* m, float* v, long long n, const float* lr_t_dev,
                          float beta1, float beta2, float eps, float grad_scale, void* shadow_bf16,
                          long long shadow_n, void* stream) {
  SLAMHIP_REQUIRE(param && grad && m && v && lr_t_dev && n >= 0, SLAMHIP_E_ARG, "adam_step_dev: bad argument");
  SLAMHIP_REQUIRE(shadow_n >= 0 && shadow_n <= n && (shadow_bf16 || shadow_n == 0), SLAMHIP_E_ARG,
                  "adam_step_dev: shadow of %lld elements for %lld parameters", shadow_n, n);
  if (n == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(adam_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, n, 0.0f,
                     beta1, beta2, eps, grad_scale, (bf16*)shadow_bf16, shadow_n, lr_t_dev);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_sgd_step(float* param, const float* grad, long long n, float lr, float grad_scale, void* stream) {
  SLAMHIP_REQUIRE(param && grad && n >= 0, SLAMHIP_E_ARG, "sgd_step: bad argument");
  if (n == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(sgd_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, n, lr, grad_scale);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_rmsprop_step(float* param, const float* grad, float* ms, long long n, float lr, float decay, float eps,
                         float grad_scale, void* stream) {
  SLAMHIP_REQUIRE(param && grad && ms && n >= 0, SLAMHIP_E_ARG, "rmsprop_step: bad argument");
  if (n == 0) return SLAMHIP_OK;
  hipLaunchKernelGGL(rmsprop_kernel, dim3(blocks_for(n)), dim3(256), 0, (hipStream_t)stream, param, grad, ms, n, lr,
                     decay, eps, grad_scale);
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

int slamhip_zero_fill(void* ptr, size_t bytes, void* stream) {
  SLAMHIP_REQUIRE(ptr || bytes == 0, SLAMHIP_E_ARG, "zero_fill: null pointer");
  SLAMHIP_HIP(zero_async(ptr, bytes, (hipStream_t)stream));
  return SLAMHIP_OK;
}

int slamhip_caller_clears_accumulators(int on) {
  const int before = t_caller_clears;
  t_caller_clears = on ? 1 : 0;
  return before;
}

int slamhip_deterministic(void) { return deterministic_mode() ? 1 : 0; }

}  // extern "C"
