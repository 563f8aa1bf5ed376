// Fused frame-pair kernel for gfx950 (MI355X): back-projection, rigid transform,
// re-projection, in-image predicates, bilinear warp of the neighbour frame and the
// residual / mask reductions in one pass over the native uint16 depth / uint8 colour
// planes, nothing materialised.
//
// Arithmetic contract (follows slam/preprocess/test_import.py:148-176): masks, counts
// and the pixel indices that choose the bilinear taps equal what the reference's
// float64 operation order gives, bit for bit.  The kernel gets there in two phases:
//   1  a cheap float64 evaluation of the same projection (per-column / per-row
//      constants hoisted, one reciprocal per pixel) whose distance from the
//      reference's value is bounded rigorously; every decision the reference takes
//      compares a coordinate with an integer (0, W-1, W, H-1, H, or floor), so a
//      pixel whose cheap coordinates are farther than the bound from every integer
//      takes the same decisions and is finished in this phase;
//   2  every other pixel (within the bound of an integer, outside the staged window,
//      behind the camera, non-finite) is redone in the reference's exact order with
//      IEEE divisions (geom_device.h), typically a few pixels in 10^5.
// Colour and depth interpolation runs in float32 (1e-4 relative to the oracle).
//
// The kernel is HBM-streaming by its bytes (10 B per pixel pair) but bound by vector
// instruction issue; nothing here is GEMM-shaped, so no MFMA.  This file is built
// with -ffp-contract=off: no product/sum pair is fused except where fma() is written.
#include <atomic>
#include <string>

#include "geom_device.h"
#include "zero_fill.h"

namespace slamhip {

// ---------------------------------------------------------------------------
// tiling
// ---------------------------------------------------------------------------
// One 256-thread workgroup owns a TILE_W x TH tile of anchor pixels.  Wave w owns the
// 64 columns (w & 1) * 64 .. +63 of rows (w >> 1) * TH/2 .. +TH/2-1: a lane walks down
// one image column, so the 64 lanes of a wave read 64 adjacent anchor pixels at every
// step (coalesced) and gather from adjacent LDS cells (no bank conflicts for the smooth
// flow of small camera motions), and everything that depends on the column alone is a
// per-lane constant.  Per neighbour frame:
//   A  wave 0 predicts the window of the neighbour frame the tile's taps fall into
//      (the tile's corners projected in float32 at the nearest and farthest depth);
//   B  all waves stage that window into LDS as 8-byte cells {R | G << 16, B | raw depth << 16}
//      (colour bytes read as binary16 subnormals) with 16-byte / 8-byte global loads;
//   C  phase 1 and phase 2 above; a tap outside the window (large parallax, a window
//      cut at WIN_PITCH x WIN_ROWS) is gathered straight from global memory - with the
//      certified coordinates of phase 1 where the pixel has them (inside the same row loop: a
//      tap pair in two vector loads, into the registers of the lane's unused window cells, blended
//      with the window lanes), in phase 2 otherwise - so the prediction never affects results.
// Predicates are 64-bit lane masks in scalar registers (one vector compare each,
// combined and counted on the scalar unit).  Reduction: sums by DPP row reductions,
// the four waves through LDS, then one float64 atomic instruction per workgroup (eight lanes:
// the four sums and the four counts) into the accumulators of the caller's workspace.  The workgroup that draws
// the last ticket of an (image, neighbour) pair moves the totals to the outputs with
// atomic exchanges, which also leaves the workspace zeroed for the next call: no memset
// and no second kernel on the stream.
constexpr int TILE_W = 128;
constexpr int CHUNK = 8;                        // staging granule in pixels (16 B of depth)
constexpr int WIN_PITCH = 160;                  // LDS window row pitch in cells (multiple of CHUNK)
// window rows for a tile of th rows: th + 8, one less where that lets four workgroups share a CU's 160 KiB
constexpr int win_rows(int th) { return th == 24 ? 31 : th + 8; }
// Staging rounds of a workgroup of tpb threads: with cx chunks per window row a round covers tpb / cx whole rows
// (thread t keeps chunk column t % cx in every round), so `rows` rows take at most this many rounds for any cx.
constexpr int stage_rounds(int tpb, int rows) {
  int m = 1;
  for (int cx = 1; cx <= WIN_PITCH / CHUNK; ++cx) {
    const int rpp = tpb / cx, n = (rows + rpp - 1) / rpp;
    if (n > m) m = n;
  }
  return m;
}

// workspace record of one (image, neighbour) pair: 128 bytes.  The counts are accumulated as float64 like the
// sums (exact below 2^53), so that a workgroup adds all eight values with ONE atomic instruction: what a
// workgroup pays at its end is the wait for its atomics before it draws its ticket, and two instructions to the
// record's line take 1.7 us longer per 64-pair launch than one.  (Drawing the ticket as a ninth lane of that
// instruction, with a second counter that tells the publishing workgroup when every add has been performed, is
// no faster: 65.6 against 65.1 us.)
struct PairAcc {
  double vals[SLAMHIP_NSUMS + SLAMHIP_NCOUNTS];   // sums, then counts
  unsigned int ticket;
  unsigned int pad[15];
};
static_assert(sizeof(PairAcc) == 128, "PairAcc must stay 128 bytes");

#ifdef SLAMHIP_AUDIT
// Diagnostic build only (-DSLAMHIP_AUDIT, tools/audit_certificate.py): every pixel that phase 1 certifies is
// projected a second time with the reference-order functions of geom_device.h, and every decision phase 1 takes
// for it is compared with the exact one.
//   [0] pixels inside the image            [1] pixels phase 1 certified       [2] pixels left to phase 2
//   [3] certified: "inside [0,W]x[0,H]" differs   [4] certified: "taps exist" differs
//   [5] certified with taps: floor(u) or floor(v) differs           [6] "x, y, z all non-zero" differs
//   [7] certified pixels whose taps were gathered from outside the window (a subset of [1])
__device__ unsigned long long g_audit[8];
__device__ __forceinline__ void audit_add(int i, unsigned long long m) {
  if (m != 0ull && __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)
    atomicAdd(&g_audit[i], (unsigned long long)__popcll(m));
}
#endif

struct FusedArgs {   // everything but the pointers (those are __restrict__ kernel parameters of their own)
  Cam cam;
  // host-rounded quotients used by phase 1 only (any rounding is covered by the bound)
  double fx_fy, fy_fx, inv_fac, inv_fx_fac, inv_fy_fac;
  float axm, aym;      // max |col - cx| / fx and max |row - cy| / fy over the image
  float extent;        // W + H + |cx| + |cy| + 8: bound of |u - cx|, |v - cy| for pixels near the image
  int B, k, H, W, tiles_x, tiles_y;
  unsigned int inv_k;  // floor(2^32 / k) + 1 for k > 1 (y / k = mulhi(y, inv_k) for y < 2^16), 0 for k == 1
  int stage_vec;       // the window can be staged with 16-byte / 8-byte loads
  int anchor_vec;      // the anchor colour can be read with one 8-byte load per pixel
  int cert_ok;         // the constants are mid-range: phase 1 may certify pixels
  int cx_col, cy_row;  // the column / row with col - cx == 0 / row - cy == 0, or -1
  unsigned long long* stamps;   // diagnostic builds only
};

// DPP row/wave reductions: the result is valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int old, int v) {
  return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, 0xF, false);
}
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}
// `old` (what masked-out rows receive) is the identity of the operation, which
// lets the compiler fold each step into one DPP-modified ALU instruction.
__device__ __forceinline__ float wave_sum63(float v) {
  v += dpp_f<0xB1, 0xF>(0.0f, v);         // quad_perm [1,0,3,2]
  v += dpp_f<0x4E, 0xF>(0.0f, v);         // quad_perm [2,3,0,1]
  v += dpp_f<0x141, 0xF>(0.0f, v);        // row_half_mirror
  v += dpp_f<0x140, 0xF>(0.0f, v);        // row_mirror
  v += dpp_f<0x142, 0xA>(0.0f, v);        // row_bcast15 into rows 1,3
  v += dpp_f<0x143, 0xC>(0.0f, v);        // row_bcast31 into rows 2,3
  return v;
}

__device__ __forceinline__ unsigned int wave_isum63(unsigned int x) {
  int v = (int)x;
  v += dpp_i<0xB1, 0xF>(0, v);
  v += dpp_i<0x4E, 0xF>(0, v);
  v += dpp_i<0x141, 0xF>(0, v);
  v += dpp_i<0x140, 0xF>(0, v);
  v += dpp_i<0x142, 0xA>(0, v);
  v += dpp_i<0x143, 0xC>(0, v);
  return (unsigned int)v;
}

// LDS cell of one neighbour-frame pixel (8 bytes):
//   x = R | G << 16,  y = B | raw_depth << 16
// A byte n in the low bits of a binary16 is the subnormal n * 2^-24: building a cell is byte
// shuffling, no conversion and no exponent to put in.  The colour arithmetic runs on these scaled
// values (subnormals are kept: the kernels run with denormal mode 3 for 16-, 32- and 64-bit operands):
// colour differences are exact packed-half subtractions, each interpolation step is one
// mixed-precision fma, the anchor colour is scaled the same way, and the two photometric sums
// are multiplied by 2^24 and 2^48 once per wave and neighbour.
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef unsigned short ushort2_t __attribute__((ext_vector_type(2)));
constexpr float COLOUR_UNSCALE = 0x1p24f;   // a colour byte read as binary16 is the byte times 2^-24

__device__ __forceinline__ half2_t as_half2(unsigned int v) { return __builtin_bit_cast(half2_t, v); }
__device__ __forceinline__ uint2 make_cell(unsigned int r, unsigned int g, unsigned int b, unsigned int depth) {
  uint2 c;
  c.x = r | (g << 16);
  c.y = b | (depth << 16);
  return c;
}
__device__ __forceinline__ unsigned int pk_min_u16(unsigned int a, unsigned int b) {
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_min(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
__device__ __forceinline__ unsigned int pk_max_u16(unsigned int a, unsigned int b) {
  return __builtin_bit_cast(unsigned int, __builtin_elementwise_max(__builtin_bit_cast(ushort2_t, a), __builtin_bit_cast(ushort2_t, b)));
}
// packed (y << 16 | x) min / max over the first eight lanes, valid in lanes 0..7
__device__ __forceinline__ unsigned int oct_pk_min(unsigned int v) {
  v = pk_min_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(-1, (int)v));
  return pk_min_u16(v, (unsigned int)dpp_i<0x141, 0xF>(-1, (int)v));
}
__device__ __forceinline__ unsigned int oct_pk_max(unsigned int v) {
  v = pk_max_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(0, (int)v));
  return pk_max_u16(v, (unsigned int)dpp_i<0x141, 0xF>(0, (int)v));
}
// packed (y << 16 | x) min / max over the wave, valid in lane 63
__device__ __forceinline__ unsigned int wave_pk_min63(unsigned int v) {
  v = pk_min_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x141, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x140, 0xF>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x142, 0xA>(-1, (int)v));
  v = pk_min_u16(v, (unsigned int)dpp_i<0x143, 0xC>(-1, (int)v));
  return v;
}
__device__ __forceinline__ unsigned int wave_pk_max63(unsigned int v) {
  v = pk_max_u16(v, (unsigned int)dpp_i<0xB1, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x4E, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x141, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x140, 0xF>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x142, 0xA>(0, (int)v));
  v = pk_max_u16(v, (unsigned int)dpp_i<0x143, 0xC>(0, (int)v));
  return v;
}

struct Acc {   // per-thread partial sums of one neighbour frame
  float ap, sp;    // photometric: |r| and r^2 on the cells' scale (2^-24 and 2^-48 of the sums)
  double ag, sg;   // geometric: the residual is formed in float64 and is summed there (no conversion per pixel)
};

struct Cells {  // the four taps of one pixel
  uint2 a, b, c, e;
};

__device__ __forceinline__ uint2 split64(unsigned long long v) { return make_uint2((unsigned int)v, (unsigned int)(v >> 32)); }
// the 2 x 2 cells at window index idx, as 8-byte LDS reads
__device__ __forceinline__ Cells read_cells(const uint2* win, unsigned int idx) {
  const unsigned long long* c = reinterpret_cast<const unsigned long long*>(win) + idx;
  Cells t;
  t.a = split64(c[0]); t.b = split64(c[1]); t.c = split64(c[WIN_PITCH]); t.e = split64(c[WIN_PITCH + 1]);
  return t;
}

__device__ __forceinline__ bool taps_nonzero(const Cells& t) {  // all four raw depths != 0
  return min(min(t.a.y, t.b.y), min(t.c.y, t.e.y)) >= 0x10000u;
}

// Bilinear blend of one pixel and its contribution to the sums, by the four tap weights
//   w11 = fu fv, w10 = fu - w11, w01 = fv - w11, w00 = (1 - fu) - w01
// (each exact to an ulp of float32, their sum 1 to 2^-23): warped - anchor is a chain of four mixed-precision
// fmas per colour channel that starts from the negated anchor, the cells' binary16 colours read in place; the
// depth taps are converted and blended the same way.  The deviation from the oracle's unfused float32
// lerps (top row, bottom row, vertical) is a few 1e-7.  a_rg / a_b: the anchor's colour in the cells'
// form, bytes R | G << 16 and B (binary16 subnormals, the colour times 2^-24); acc.ap and acc.sp therefore
// collect 2^-24 and 2^-48 of the photometric sums.  z2: the transformed depth (float64).  geo: the pixel also
// has four non-zero depth taps (taps_nonzero; the caller counts these pixels from the lane mask, outside
// the divergent region, so that the count stays a scalar).  Runs only on lanes whose pixel warps (warp_valid).
template <bool UNIT>
__device__ __forceinline__ void blend_accumulate(Acc& acc, const bool geo, const Cells& t, float fu, float fv,
                                                 float inv_factor_f, unsigned int a_rg, unsigned int a_b, double z2) {
  const float w11 = fu * fv, w10 = fu - w11, w01 = fv - w11, w00 = (1.0f - fu) - w01;
  const half2_t aRG = as_half2(t.a.x), bRG = as_half2(t.b.x), cRG = as_half2(t.c.x), eRG = as_half2(t.e.x);
  const _Float16 aB = as_half2(t.a.y).x, bB = as_half2(t.b.y).x, cB = as_half2(t.c.y).x, eB = as_half2(t.e.y).x;
  const half2_t nRG = as_half2(a_rg);
  const _Float16 nB = as_half2(a_b).x;
  // warped - anchor (the sign is irrelevant to |r| and r^2)
  float rr = __builtin_fmaf(w00, (float)aRG.x, -(float)nRG.x), rg = __builtin_fmaf(w00, (float)aRG.y, -(float)nRG.y);
  float rb = __builtin_fmaf(w00, (float)aB, -(float)nB);
  rr = __builtin_fmaf(w10, (float)bRG.x, rr); rg = __builtin_fmaf(w10, (float)bRG.y, rg); rb = __builtin_fmaf(w10, (float)bB, rb);
  rr = __builtin_fmaf(w01, (float)cRG.x, rr); rg = __builtin_fmaf(w01, (float)cRG.y, rg); rb = __builtin_fmaf(w01, (float)cB, rb);
  rr = __builtin_fmaf(w11, (float)eRG.x, rr); rg = __builtin_fmaf(w11, (float)eRG.y, rg); rb = __builtin_fmaf(w11, (float)eB, rb);
  acc.ap = ((acc.ap + fabsf(rr)) + fabsf(rg)) + fabsf(rb);
  acc.sp = __builtin_fmaf(rb, rb, __builtin_fmaf(rg, rg, __builtin_fmaf(rr, rr, acc.sp)));
  if (geo) {   // geo_valid: the depth channel and its residual
    const float da = (float)(t.a.y >> 16), db = (float)(t.b.y >> 16), dc = (float)(t.c.y >> 16), de = (float)(t.e.y >> 16);
    const float w3 = __builtin_fmaf(w11, de, __builtin_fmaf(w01, dc, __builtin_fmaf(w10, db, w00 * da)));
    const float wd = UNIT ? w3 : w3 * inv_factor_f;
    // in float64: test_import's d is float64 and may be far larger than the residual
    const double g = (double)wd - z2;
    acc.ag += fabs(g);
    acc.sg = fma(g, g, acc.sg);
  }
}

__device__ __forceinline__ uint2 global_cell(const uint16_t* __restrict__ d2, const uint8_t* __restrict__ c2, int W,
                                             int x, int y) {
  const int o = y * W + x;
  return make_cell(c2[3 * o], c2[3 * o + 1], c2[3 * o + 2], d2[o]);
}

// The taps (x, y), (x + 1, y) of a pixel with x + 1 < W, straight from the neighbour's planes: exactly the ten bytes
// that hold them (and the two after), in two loads at whatever alignment the pixel has (byte-aligned vector loads
// are legal on gfx9 under the HSA runtime's memory configuration).
struct __attribute__((packed, aligned(1))) U32u { unsigned int v; };
struct __attribute__((packed, aligned(1))) U64u { unsigned int x, y; };
// two raw depths; eight bytes of which bytes sh .. sh + 5 are R0 G0 B0 R1 G1 B1 (the load is pulled back by sh <= 2
// bytes at the plane's last two pixels so that it stays inside the plane)
struct TapPair { unsigned int dd, qx, qy, sh; };
__device__ __forceinline__ TapPair load_tap_pair(const uint16_t* __restrict__ d2, const uint8_t* __restrict__ c2,
                                                 unsigned int o, unsigned int plane_bytes) {
  TapPair t;
  t.dd = reinterpret_cast<const U32u*>(reinterpret_cast<const char*>(d2) + (size_t)(2u * o))->v;
  const unsigned int at = 3u * o, base = min(at, plane_bytes - 8u);
  t.sh = at - base;
  const U64u q = *reinterpret_cast<const U64u*>(reinterpret_cast<const char*>(c2) + (size_t)base);
  t.qx = q.x; t.qy = q.y;
  return t;
}
__device__ __forceinline__ void tap_pair_cells(const TapPair& t, uint2& left, uint2& right) {
  const unsigned int ca = __builtin_amdgcn_alignbyte(t.qy, t.qx, t.sh);   // R0 G0 B0 R1
  const unsigned int cb = t.qy >> (8u * t.sh);                            // G1 B1 . .
  left.x = __builtin_amdgcn_perm(0u, ca, 0x0c010c00u);      // R0 | G0 << 16
  left.y = __builtin_amdgcn_perm(t.dd, ca, 0x05040c02u);    // B0 | depth0 << 16
  right.x = __builtin_amdgcn_perm(cb, ca, 0x0c040c03u);     // R1 | G1 << 16
  right.y = __builtin_amdgcn_perm(t.dd, cb, 0x07060c01u);   // B1 | depth1 << 16
}

// Projection of one anchor pixel in float32, for window prediction only.
__device__ __forceinline__ void project_f32(const Cam& c, const Pose& p, float u0, float v0, float z, float& u,
                                            float& v, float& z2) {
  const float x = (u0 - (float)c.cx) * z * __builtin_amdgcn_rcpf((float)c.fx);
  const float y = (v0 - (float)c.cy) * z * __builtin_amdgcn_rcpf((float)c.fy);
  const float x2 = x * (float)p.r00 + y * (float)p.r10 + z * (float)p.r20 + (float)p.t0;
  const float y2 = x * (float)p.r01 + y * (float)p.r11 + z * (float)p.r21 + (float)p.t1;
  z2 = x * (float)p.r02 + y * (float)p.r12 + z * (float)p.r22 + (float)p.t2;
  const float rz = __builtin_amdgcn_rcpf(z2);
  u = x2 * (float)c.fx * rz + (float)c.cx;
  v = y2 * (float)c.fy * rz + (float)c.cy;
}

// Diagnostic build only (-DSLAMHIP_STAMPS): the shader clock at the phase boundaries of every workgroup's
// wave 0, written to a buffer of their own that nothing else reads.
#ifdef SLAMHIP_STAMPS
static unsigned long long* g_stamps = nullptr;
#define SLAMHIP_STAMP(i) do { if (a.stamps && wave == 0 && lane == 0) a.stamps[(size_t)wg_linear * 16 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define SLAMHIP_STAMP(i) do { } while (0)
#endif

struct __attribute__((aligned(4))) Dword2 {   // eight bytes at a 4-byte-aligned address
  unsigned int x, y;
};

// The kernel's parameter list as it lies in the kernel-argument segment.  The single-neighbour kernel reads
// what its rare and late parts need (the exact phase: camera, pose, plane pointers; the publication: workspace and
// outputs) from there at the point of use, through a pointer the optimiser cannot see through: kept from the
// kernel's start they are some 50 scalar registers that live across the row loop, and the ones beyond the
// register file are parked in vector-register lanes, one vector instruction per move.
struct FusedKernArgs {
  FusedArgs a;
  const uint16_t* depth1; const uint8_t* rgb1; const uint16_t* depth2; const uint8_t* rgb2;
  const double* rel; PairAcc* ws; double* sums; unsigned int* counts;
};
static_assert(offsetof(FusedKernArgs, depth1) == sizeof(FusedArgs) && sizeof(FusedArgs) % 8 == 0, "kernel-argument layout");
typedef const __attribute__((address_space(4))) FusedKernArgs* KernArgPtr;
typedef const __attribute__((address_space(4))) double* ConstF64Ptr;

__device__ __forceinline__ KernArgPtr late_kernargs() {
  KernArgPtr q = (KernArgPtr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(q));
  return q;
}

// the pose through the constant address space (the matrices are not written while the kernel runs): scalar loads
__device__ __forceinline__ Pose load_pose_const(const double* m) {
  const ConstF64Ptr c = (ConstF64Ptr)(unsigned long long)m;
  Pose p;
  p.r00 = c[0]; p.r01 = c[1]; p.r02 = c[2]; p.t0 = c[3];
  p.r10 = c[4]; p.r11 = c[5]; p.r12 = c[6]; p.t1 = c[7];
  p.r20 = c[8]; p.r21 = c[9]; p.r22 = c[10]; p.t2 = c[11];
  return p;
}

// waves per SIMD the register allocation must allow: what the LDS windows of TH-row tiles leave room for
constexpr int fused_waves(int th, int nw) { const int wg = 163840 / (TILE_W + 32) / 8 / win_rows(th); return (wg * nw + 3) / 4 > 8 ? 8 : (wg * nw + 3) / 4; }

// Constants of phase 1 that depend on the pair alone (wave-uniform): see the phase's comments for their meaning.
struct PairConsts {
  double kAc, kAr, kA0, kBc, kBr, kB0, kCc, kCr, kC0, t0p, t1p, t2, dz;
  float half_eps;
  int cert;
};

template <int TH, int RPW, bool UNIT, bool K1>
__global__ void __launch_bounds__(128 * (TH / RPW), fused_waves(TH, 2 * (TH / RPW)))
pair_loss_fused_kernel(const FusedArgs a, const uint16_t* __restrict__ depth1, const uint8_t* __restrict__ rgb1,
                       const uint16_t* __restrict__ depth2, const uint8_t* __restrict__ rgb2,
                       const double* __restrict__ rel, PairAcc* __restrict__ ws, double* __restrict__ sums,
                       unsigned int* __restrict__ counts) {
  // RPW: rows a wave (and every lane of it) walks; the tile's TH / RPW row groups x 2 column halves = NW waves
  constexpr int NW = 2 * (TH / RPW), TPB = 64 * NW;
  static_assert(TH % RPW == 0 && NW >= 2 && NW <= 4, "two or four waves per tile");
  constexpr int WIN_ROWS = win_rows(TH);
  static_assert(RPW >= 1 && RPW <= 32, "one bit per row in a 32-bit mask");
  __shared__ uint2 win[WIN_PITCH * WIN_ROWS];
  __shared__ unsigned int s_rng[NW][2];
  __shared__ int s_winrec[6];
  __shared__ float s_sum[NW][4];
  __shared__ unsigned int s_cnt[NW][4];
  __shared__ PairConsts s_pc;

  const int H = a.H, W = a.W, k = a.k;
  const Cam cam = a.cam;
  // Grid: x = 8 * tiles_x (the low three bits are the XCD: workgroups are dealt to the XCDs round-robin by their
  // linear index, and the x extent is a multiple of 8), y = tiles_y * k (K1) or tiles_y, z = groups of eight anchors.
  // Whole images go to one XCD, so a tile's halo rows / columns hit in its L2; with K1 a workgroup takes one tile
  // of ONE (anchor, neighbour) pair, and the k pairs of an anchor tile are 8 * tiles_x workgroups apart on the same
  // XCD: all but the first find the anchor tile in L2.  Otherwise one tile of an anchor for all its neighbours in
  // turn.  Everything here is scalar arithmetic (the division by k is a multiplication by its host-made reciprocal).
  const int group = blockIdx.x & 7;
  const int tx = blockIdx.x >> 3;
  const unsigned int y = blockIdx.y;
  const int ty = (K1 && a.inv_k) ? (int)__umulhi(y, a.inv_k) : (int)y;
  const int j0 = K1 ? (int)y - ty * k : 0;
  const int b = (int)blockIdx.z * 8 + group;
  if (b >= a.B) return;
#ifdef SLAMHIP_STAMPS
  const size_t wg_linear = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z);
#endif
  const int tiles = a.tiles_x * a.tiles_y;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // known to be wave-uniform
  SLAMHIP_STAMP(0);   // start
#ifdef SLAMHIP_STAMPS
  if (a.stamps && wave == 0 && lane == 0) a.stamps[(size_t)wg_linear * 16 + 13] = __builtin_amdgcn_s_memrealtime();
#endif
  const long long HW = (long long)H * W;
  const int col = tx * TILE_W + (wave & 1) * 64 + lane;
  const int row0 = ty * TH + (wave >> 1) * RPW;
  const int colc = min(col, W - 1);          // a column outside the image has no bit in the masks below
  const uint16_t* d1b = depth1 + (long long)b * HW;
  const uint8_t* c1b = rgb1 + 3 * (long long)b * HW;

  // rows of this wave inside the image, and of those the rows with row - cy != 0
  // (closed forms: as a chain of per-row compares the masks are rebuilt where they are used, and the twelve row
  // numbers and compare results live in scalar registers until then)
  const int n_in = min(max(H - row0, 0), RPW);
  unsigned int rows_in = (n_in >= 32) ? 0xffffffffu : ((1u << n_in) - 1u);
  const unsigned int rz = (unsigned int)(a.cy_row - row0);
  unsigned int rows_nz = rows_in & ~((rz < (unsigned int)RPW) ? (1u << rz) : 0u);

  // the first neighbour's pose is asked for before anything else: the window prediction waits for it
  Pose p_next = load_pose(rel + 16 * ((long long)b * k + j0));
#ifdef SLAMHIP_EXP_POSEWAIT   // diagnostic: how long the pose takes by itself
  asm volatile("s_waitcnt lgkmcnt(0)" :: "s"(p_next.t0), "s"(p_next.r00));
  SLAMHIP_STAMP(15);
#endif

  // ---- anchor pixels of this lane: raw depth and the eight bytes that hold the colour, read once per tile;
  // nothing here waits for the loads: they stay in flight behind the window prediction and the staging ----
  unsigned int a_d[RPW];
  Dword2 a_q[RPW];
  // byte selectors that turn a_q into the cells' form of the colour: {R, 0, G, 0} and {B, 0, 0, 0} (a byte read as
  // binary16); the eight loaded bytes stay as they are until the pixel's row needs them
  unsigned int a_sel_rg, a_sel_b;
  {
    const unsigned int d_off = 2u * (unsigned int)colc;
    if (a.anchor_vec) {
      // W % 4 == 0 and a 4-byte aligned plane: 3 * (row * W + col) has the alignment of 3 * col in every
      // row, so one 8-byte load at a fixed per-lane offset holds the pixel's three bytes (the offset is
      // pulled back at the row's end so that the load stays inside the row)
      const unsigned int L = 3u * (unsigned int)colc;
      const unsigned int La = min(L & ~3u, 3u * (unsigned int)W - 8u);
      const unsigned int s = L - La;   // 0..5
      a_sel_rg = s | ((s + 1u) << 16) | 0x0c000c00u;
      a_sel_b = (s + 2u) | 0x0c0c0c00u;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        // (32-bit row offsets: an image has at most 2^29 pixels; scalar base + per-lane 32-bit offset)
        const unsigned int rw = (unsigned int)min(row0 + r, H - 1) * (unsigned int)W;
        const char* drow = reinterpret_cast<const char*>(d1b) + (size_t)(2u * rw);
        const char* crow = reinterpret_cast<const char*>(c1b) + (size_t)(3u * rw);
#ifdef SLAMHIP_ABL_HALF_ANCHORS   // timing ablation only: a quarter of the anchor loads
        if (r & 3) { a_d[r] = 1000u + r; a_q[r].x = a_q[r].y = 0x10203040u; continue; }
#endif
#ifdef SLAMHIP_ABL_ARITH_ONLY     // timing ablation only: no global loads at all (with SLAMHIP_ABL_NO_STAGE)
        { a_d[r] = 5000u + 3u * (unsigned int)r + (unsigned int)lane; a_q[r].x = 0x10203040u + (unsigned int)lane; a_q[r].y = 0x50607080u; continue; }
#endif
        a_d[r] = *reinterpret_cast<const uint16_t*>(drow + d_off);
        a_q[r] = *reinterpret_cast<const Dword2*>(crow + La);
      }
    } else {
      const unsigned int c_off = 3u * (unsigned int)colc;
      a_sel_rg = 0x0c010c00u;
      a_sel_b = 0x0c0c0c02u;
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const unsigned int rw = (unsigned int)min(row0 + r, H - 1) * (unsigned int)W;
        const char* drow = reinterpret_cast<const char*>(d1b) + (size_t)(2u * rw);
        const uint8_t* cp = c1b + (size_t)(3u * rw) + c_off;
        a_d[r] = *reinterpret_cast<const uint16_t*>(drow + d_off);
        a_q[r].x = (unsigned int)cp[0] | ((unsigned int)cp[1] << 8) | ((unsigned int)cp[2] << 16);
        a_q[r].y = 0u;
      }
    }
  }
  // (wave 0 asking for its anchors only behind its staging loads, so that it can start on the window prediction
  // at once, gives the anchor registers two definition sites: 128 vector registers and 212 bytes of scratch)
  SLAMHIP_STAMP(1);   // anchor loads issued
  const float factor_f = (float)cam.factor;
  const float inv_factor_f = 1.0f / factor_f;
  const double ucx = (double)col - cam.cx;
  const double vcy0 = (double)row0 - cam.cy;

  // range of non-zero anchor depth in the tile: found on first need (see below)
  unsigned int zmn = 0xffffu, zmx = 0u;
  bool have_range = false;

  // ---- certificate of phase 1 (all values wave-uniform) ----
  // With S = 65535/factor * (axm |r0j| + aym |r1j| + |r2j|) + |tj| (the largest sum of magnitudes in
  // the transform of coordinate j), the reference's x', y', z' and the ones of phase 1 each differ from
  // the real-number value by at most 8 resp. 24 roundings of 2^-53 S (constants, the per-row steps,
  // the fmas).  For z' above delta_z = 2^-16 Sz the projected coordinate therefore moves by at most
  //   32 * 2^-53 * 2^17 * (fx Sx / Sz + |u - cx|)  =  2^-31 * (...)  <  eps = 2^-30 * ((fx Sx + fy Sy) / Sz + extent)
  // (the reciprocal's 2^-48 and the final roundings are covered by the slack of 2), and eps_f, the
  // threshold applied to the float32 fractions, is twice that (which also covers pixels up to twice
  // `extent` away from the principal point) and at least 2^-20.
  // The single-neighbour kernel computes them once per workgroup (wave 1, while wave 0 predicts the window) and
  // hands them over in LDS: they are some 150 vector instructions that every wave would otherwise repeat.
  auto pair_consts = [&](const Pose& q) -> PairConsts {
    PairConsts c;
    const float zmaxf = 65535.0f * (float)a.inv_fac;
    const float Sx = zmaxf * (a.axm * fabsf((float)q.r00) + a.aym * fabsf((float)q.r10) + fabsf((float)q.r20)) + fabsf((float)q.t0);
    const float Sy = zmaxf * (a.axm * fabsf((float)q.r01) + a.aym * fabsf((float)q.r11) + fabsf((float)q.r21)) + fabsf((float)q.t1);
    const float Sz = zmaxf * (a.axm * fabsf((float)q.r02) + a.aym * fabsf((float)q.r12) + fabsf((float)q.r22)) + fabsf((float)q.t2);
    const float delta_z = Sz * 0x1p-16f;
    const float ratio = ((float)cam.fx * Sx + (float)cam.fy * Sy) / Sz;
    const float eps = 0x1p-30f * (ratio + a.extent);
    const float eps_f = fmaxf(2.0f * eps, 0x1p-20f);
    c.cert = (a.cert_ok != 0) && (eps_f <= 0x1p-7f) && (delta_z > 0.0f) && (delta_z < 3.0e38f);   // NaN fails
    c.half_eps = 0.5f - eps_f;
    c.dz = (double)delta_z;
    // u - wx0 + 1 = (z A + t0') / (z C + t2) + (cx - wx0 + 1) with
    //   A = (col-cx) r00 + (row-cy) (fx/fy) r10 + fx r20,  C = ((col-cx)/fx) r02 + ((row-cy)/fy) r12 + r22
    // (all divided by factor: z is the raw depth), likewise v; A, B, C advance by a constant per row
    c.kAc = q.r00 * a.inv_fac; c.kAr = (a.fx_fy * q.r10) * a.inv_fac; c.kA0 = (cam.fx * q.r20) * a.inv_fac;
    c.kBc = (a.fy_fx * q.r01) * a.inv_fac; c.kBr = q.r11 * a.inv_fac; c.kB0 = (cam.fy * q.r21) * a.inv_fac;
    c.kCc = q.r02 * a.inv_fx_fac; c.kCr = q.r12 * a.inv_fy_fac; c.kC0 = q.r22 * a.inv_fac;
    c.t0p = cam.fx * q.t0; c.t1p = cam.fy * q.t1; c.t2 = q.t2;
    return c;
  };

  // (K1: straight-line code for the one neighbour j0; the other form walks the anchor's k neighbours)
  const int j_end = K1 ? j0 + 1 : k;
#pragma unroll 1
  for (int j = j0; j < j_end; ++j) {
    const long long bj = (long long)b * k + j;
    // (the pose of a later neighbour is loaded here, at its turn: held in registers across the rows of the
    // neighbour before it, its 24 scalar registers cost more in spill moves than the 0.3 us it waits now)
    Pose p = p_next;
    if (!K1 && j > 0) p = load_pose(rel + 16 * bj);
    const uint16_t* d2 = depth2 + bj * HW;
    const uint8_t* c2 = rgb2 + 3 * bj * HW;

    // ---- predicted tap window: the tile's corners at the nearest and farthest depth ----
    // For a fixed depth the tile maps to a convex quadrilateral, and a pixel moves
    // along a segment as its depth varies, so the eight projected points bound
    // every tap.  The window is only a cache: a tap outside it is gathered from
    // global memory, so the prediction (float32, with a margin) affects speed only.
    // When the translation moves a point at 256 raw depth units by less than half a
    // pixel, the window does not depend on depth and the tile's depth range is
    // not needed (a nearer pixel may miss the window and take the slow path).
    const float shift_px = fmaxf((float)cam.fx, (float)cam.fy) * factor_f * (1.0f / 256.0f) *
                           (fabsf((float)p.t0) + fabsf((float)p.t1) + fabsf((float)p.t2));
    const bool depth_free = shift_px < 0.5f;   // NaN: false
#ifdef SLAMHIP_STAMPS
    if (__builtin_amdgcn_readfirstlane((int)depth_free) >= 0) SLAMHIP_STAMP(12);   // pose arrived
#endif
    if (!depth_free && !have_range) {
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        const bool inside = ((rows_in >> r) & 1u) && col < W;
        const unsigned int dr = inside ? a_d[r] : 0u;
        zmn = min(zmn, dr ? dr : 0xffffu);
        zmx = max(zmx, dr);
      }
      const unsigned int pk = wave_pk_min63(zmn | 0xffff0000u), px = wave_pk_max63(zmx);
      if (lane == 63) { s_rng[wave][0] = pk; s_rng[wave][1] = px; }
      __syncthreads();
      zmn = 0xffffu; zmx = 0u;
#pragma unroll
      for (int w = 0; w < NW; ++w) { zmn = min(zmn, s_rng[w][0] & 0xffffu); zmx = max(zmx, s_rng[w][1] & 0xffffu); }
      have_range = true;
    }
    if (wave == 0) {
      __builtin_amdgcn_s_setprio(3);   // the other waves of the workgroup wait for this one (0.7-0.9 us per launch)
      const unsigned int z_near = depth_free ? 256u : zmn, z_far = depth_free ? 65535u : zmx;
      const bool any_depth = depth_free || zmx != 0u;
      int wx0 = 0, wy0 = 0, ncols = 0, nrows = 0;
      if (any_depth) {
        const int cx_ = (lane & 1) ? min(tx * TILE_W + TILE_W - 1, W - 1) : tx * TILE_W;
        const int cy_ = (lane & 2) ? min(ty * TH + TH - 1, H - 1) : ty * TH;
        const float zc = (float)((lane & 4) ? z_far : z_near) * __builtin_amdgcn_rcpf(factor_f);
        float pu, pv, pz;
        project_f32(cam, p, (float)cx_, (float)cy_, zc, pu, pv, pz);
        const bool okp = (lane < 8) && (pz > 0.0f) && (fabsf(pu) < 1e6f) && (fabsf(pv) < 1e6f);   // NaN fails
        const bool bad = __builtin_amdgcn_ballot_w64((lane < 8) && !okp) != 0ull;
        const int lo_u = okp ? max((int)floorf(pu) - 2, 0) : 0x7fff, lo_v = okp ? max((int)floorf(pv) - 2, 0) : 0x7fff;
        const int hi_u = okp ? min((int)floorf(pu) + 3, W - 1) : 0, hi_v = okp ? min((int)floorf(pv) + 3, H - 1) : 0;
        const unsigned int pmin = oct_pk_min(((unsigned int)min(lo_v, 0x7fff) << 16) | (unsigned int)min(lo_u, 0xffff));
        const unsigned int pmax = oct_pk_max(((unsigned int)max(hi_v, 0) << 16) | (unsigned int)max(hi_u, 0));
        const unsigned int bmin = (unsigned int)__builtin_amdgcn_readfirstlane((int)pmin);
        const unsigned int bmax = (unsigned int)__builtin_amdgcn_readfirstlane((int)pmax);
        const int bx0 = (int)(bmin & 0xffffu), by0 = (int)(bmin >> 16), bx1 = (int)(bmax & 0xffffu), by1 = (int)(bmax >> 16);
        if (!bad && bx1 >= bx0 && by1 >= by0 && a.stage_vec) {
          wx0 = bx0 & ~(CHUNK - 1);
          wy0 = by0;
          if (bx1 + 1 - wx0 > WIN_PITCH || by1 + 1 - wy0 > WIN_ROWS) {
            // The taps' box is larger than the window (parallax over a wide depth range).  Parallax goes with 1 / z,
            // so equal steps of depth crowd together at the far end of the box: the window is laid over the
            // projections of the tile at its farthest depth and reaches from there towards the near end.
            const bool far_corner = (lane & 4) != 0;
            const unsigned int fmn = oct_pk_min(far_corner ? (((unsigned int)min(lo_v, 0x7fff) << 16) | (unsigned int)min(lo_u, 0xffff)) : 0xffffffffu);
            const unsigned int fmx = oct_pk_max(far_corner ? (((unsigned int)max(hi_v, 0) << 16) | (unsigned int)max(hi_u, 0)) : 0u);
            const unsigned int smn = (unsigned int)__builtin_amdgcn_readfirstlane((int)fmn);
            const unsigned int smx = (unsigned int)__builtin_amdgcn_readfirstlane((int)fmx);
            const int fx0 = (int)(smn & 0xffffu), fy0 = (int)(smn >> 16), fx1 = (int)(smx & 0xffffu), fy1 = (int)(smx >> 16);
            if (fx0 - bx0 > bx1 - fx1) wx0 = max(wx0, (fx1 + 1 - WIN_PITCH + (CHUNK - 1)) & ~(CHUNK - 1));
            if (fy0 - by0 > by1 - fy1) wy0 = max(wy0, fy1 + 1 - WIN_ROWS);
          }
          ncols = min(bx1 + 1 - wx0, WIN_PITCH);   // a larger box is cut: the rest goes the slow way
          nrows = min(by1 + 1 - wy0, WIN_ROWS);
          if (ncols < 2 || nrows < 2) ncols = nrows = 0;   // a tap needs two columns and two rows
        }
      }
      // chunks per window row, its 16-bit reciprocal (t / cx = (t * inv) >> 16, exact for t < 2^16 / cx) and the
      // whole rows a staging round covers: made here once instead of by every wave after the barrier
      const unsigned int cxs = (unsigned int)(ncols + CHUNK - 1) / CHUNK;
      const unsigned int inv = cxs ? 65535u / cxs + 1u : 0u;
      if (lane == 0) {
        s_winrec[0] = wx0; s_winrec[1] = wy0; s_winrec[2] = ncols; s_winrec[3] = nrows;
        s_winrec[4] = (int)inv; s_winrec[5] = (int)(((unsigned int)TPB * inv) >> 16);
      }
      __builtin_amdgcn_s_setprio(0);
    }
    if constexpr (K1) {
      if (wave == 1) {
        const PairConsts c = pair_consts(p);
        if (lane == 0) s_pc = c;
      }
    }
    SLAMHIP_STAMP(2);   // window predicted
    __syncthreads();   // also: the previous neighbour's gathers and s_sum / s_cnt reads are done
    SLAMHIP_STAMP(3);   // barrier passed
    const int wx0 = __builtin_amdgcn_readfirstlane(s_winrec[0]), wy0 = __builtin_amdgcn_readfirstlane(s_winrec[1]);
    const int ncols = __builtin_amdgcn_readfirstlane(s_winrec[2]), nrows = __builtin_amdgcn_readfirstlane(s_winrec[3]);

    // ---- stage the window: 8-byte cells {half R | half G, half B | raw depth} ----
    // A thread keeps one chunk column (8 pixels) and walks down the window: round q holds rows q * rpp .. + rpp - 1,
    // thread t the chunk (q * rpp + t / chunks_x, t % chunks_x).  Source and destination then advance by a
    // wave-uniform step per round: per-thread 32-bit offsets are made once and the planes' bases move in scalar
    // registers (a chunk's address costs no vector instruction after the first round).
    const int chunks_x = (ncols + CHUNK - 1) / CHUNK;
    const unsigned int inv_cx = (unsigned int)__builtin_amdgcn_readfirstlane(s_winrec[4]);
    const int rpp = __builtin_amdgcn_readfirstlane(s_winrec[5]);
    auto chunk_store = [&](uint2* dst, const uint4 dv, const uint2 q0, const uint2 q1, const uint2 q2) {
      const uint32_t dw[4] = {dv.x, dv.y, dv.z, dv.w};
      const uint32_t cw[7] = {q0.x, q0.y, q1.x, q1.y, q2.x, q2.y, 0u};
#pragma unroll
      for (int i = 0; i < CHUNK; ++i) {
        // bytes 3i, 3i+1, 3i+2 of the 24-byte colour group; v_perm_b32 picks bytes
        // from {hi dword (4..7), lo dword (0..3)}, selector 0x0c gives a zero byte
        const int br = 3 * i, bg = 3 * i + 1, bb = 3 * i + 2;
        const int kk = br >> 2;   // R and G lie in dwords kk, kk+1
        const unsigned int sel_rg = 0x0c000c00u | (unsigned int)(br - 4 * kk) | ((unsigned int)(bg - 4 * kk) << 16);
        const unsigned int sel_bd = 0x00000c00u | (unsigned int)(bb & 3) | ((4u + 2u * (i & 1)) << 16) | ((5u + 2u * (i & 1)) << 24);
        uint2 cell;
        cell.x = __builtin_amdgcn_perm(cw[kk + 1], cw[kk], sel_rg);
        cell.y = __builtin_amdgcn_perm(dw[i >> 1], cw[bb >> 2], sel_bd);
#ifdef SLAMHIP_ABL_LINEAR_STAGE   // timing ablation only: lane-contiguous (conflict-free) but wrong cell positions
        win[(i * 256 + (threadIdx.x & 255)) % (WIN_PITCH * WIN_ROWS)] = cell;
#else
        dst[i] = cell;
#endif
      }
    };
    // one pass: every thread has the loads of all its chunks in flight before it converts the first
    constexpr int NQ = stage_rounds(TPB, WIN_ROWS);
#ifndef SLAMHIP_ABL_NO_STAGE      // (timing ablation only: results are wrong without the staging)
    if (chunks_x > 0 && nrows > 0) {
      const unsigned int cr0 = ((unsigned int)tid * inv_cx) >> 16, cc = (unsigned int)tid - cr0 * (unsigned int)chunks_x;
      const bool mine = cr0 < (unsigned int)rpp;      // (the last tpb % chunks_x threads have no chunk)
      const unsigned int wave_r0 = ((unsigned int)(wave * 64) * inv_cx) >> 16;   // first window row of this wave's chunks
      // pixel offset of the thread's chunk in round 0 (inside the image, 8-pixel aligned; 32 bits: an image has at most 2^29 pixels)
      const unsigned int pix0 = (unsigned int)(wy0 + (int)cr0) * (unsigned int)W + (unsigned int)wx0 + cc * CHUNK;
      const unsigned int d_off = 2u * pix0, c_off = 3u * pix0;
      uint2* const dst0 = win + cr0 * WIN_PITCH + cc * CHUNK;
      const unsigned int rstep = (unsigned int)rpp * (unsigned int)W;   // pixels per round
      uint4 dvq[NQ];
      uint2 cq[NQ][3];
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if ((int)wave_r0 + q * rpp >= nrows) continue;   // (wave-uniform) none of this wave's chunks of this round exists
        if (mine && (int)cr0 + q * rpp < nrows) {
          const char* dq = reinterpret_cast<const char*>(d2) + (size_t)(2u * rstep * (unsigned int)q);
          const char* cqp = reinterpret_cast<const char*>(c2) + (size_t)(3u * rstep * (unsigned int)q);
          dvq[q] = *reinterpret_cast<const uint4*>(dq + d_off);
          const uint2* cp = reinterpret_cast<const uint2*>(cqp + c_off);
          cq[q][0] = cp[0]; cq[q][1] = cp[1]; cq[q][2] = cp[2];
        }
      }
      SLAMHIP_STAMP(4);   // staging loads issued
#ifdef SLAMHIP_STAMPS
      asm volatile("s_waitcnt vmcnt(0)");
#endif
      SLAMHIP_STAMP(5);   // ... and landed (diagnostic builds only wait here for all of them)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if ((int)wave_r0 + q * rpp >= nrows) continue;
        if (mine && (int)cr0 + q * rpp < nrows) chunk_store(dst0 + q * rpp * WIN_PITCH, dvq[q], cq[q][0], cq[q][1], cq[q][2]);
      }
    }
#endif
    SLAMHIP_STAMP(6);   // cells written
    __syncthreads();
    SLAMHIP_STAMP(7);   // barrier passed

    // a tap (du, dv) is in the window when du < lim_u and dv < lim_v (unsigned compares)
    const unsigned int lim_u = (unsigned int)max(chunks_x * CHUNK - 1, 0), lim_v = (unsigned int)max(nrows - 1, 0);

    Acc acc = {0.f, 0.f, 0.0, 0.0};
    unsigned int n_valid0 = 0, n_inb = 0, n_warp = 0, n_fast = 0;
    unsigned int n_geo = 0u;       // pixels with four non-zero depth taps (counted from the lane masks)
    // bit r: row r of this lane is left to phase 2
    unsigned int rest_bits = 0u;

    PairConsts pc;
    if constexpr (K1) pc = s_pc;     // written by wave 1 ahead of the first barrier
    else pc = pair_consts(p);
    const bool cert_u = __builtin_amdgcn_readfirstlane(pc.cert) != 0;   // the same in every lane

    // (the anchors and row masks are made opaque to the optimiser once per neighbour: otherwise everything
    // that depends on them alone is computed for all rows ahead of the neighbour loop and held in registers;
    // here, after the staging, so that the anchor loads stay in flight behind the prediction and the staging)
    unsigned int sel_rg = a_sel_rg, sel_b = a_sel_b;
    asm volatile("" : "+v"(sel_rg), "+v"(sel_b));
#pragma unroll
    for (int r = 0; r < RPW; ++r) asm volatile("" : "+v"(a_d[r]), "+v"(a_q[r].x), "+v"(a_q[r].y));
    asm volatile("" : "+s"(rows_nz), "+s"(rows_in));
    // columns of this wave inside the image, and of those the ones with col - cx != 0 (made here, at their use:
    // held from the kernel's start the two masks are parked in vector lanes across the staging)
    int colq = col;
    asm volatile("" : "+v"(colq));
    const unsigned long long col_in = __builtin_amdgcn_ballot_w64(colq < W);
    const unsigned long long col_nz = __builtin_amdgcn_ballot_w64(colq < W && colq != a.cx_col);
#if defined(SLAMHIP_ABL_NO_PHASE1) || defined(SLAMHIP_ABL_LOADS_ONLY)   // timing ablation only: results are wrong
    if (false) {
#else
    if (cert_u && nrows > 0) {
#endif
      // ---- phase 1 (constants: pair_consts above) ----
      const double kAr = pc.kAr, kBr = pc.kBr, kCr = pc.kCr;
      double A = fma(ucx, pc.kAc, fma(vcy0, kAr, pc.kA0));
      double Bq = fma(ucx, pc.kBc, fma(vcy0, kBr, pc.kB0));
      double C = fma(ucx, pc.kCc, fma(vcy0, kCr, pc.kC0));
      const double t0p = pc.t0p, t1p = pc.t1p, t2 = pc.t2;
      // Window-relative coordinates carry an integer bias of 2^16 so that the truncating conversion below IS the
      // floor for every coordinate above wx0 - 65537 (a coordinate in (-1, 0) would truncate to 0 = floor + 1
      // otherwise, and a certified pixel there whose other coordinate lies in the last row or column would be
      // counted as inside the image); below that, floor + 1 is still negative, i.e. outside.  The bias costs no
      // instruction (it sits in the two constants) and 2^-36 of absolute accuracy, far inside the certificate's bound.
      constexpr double WBIAS = 65536.0;
      const double cxw = cam.cx - (double)(wx0 - 1) + WBIAS, cyw = cam.cy - (double)(wy0 - 1) + WBIAS;
      const double dz = pc.dz;
      const float half_eps = pc.half_eps;
      // Software pipeline: row r + 1 is projected and its four cells are requested from LDS before row r
      // is blended, so that the LDS latency and the projection's dependent float64 chain overlap the
      // blend of the row before (a row's code is several basic blocks: the compiler cannot do this).
      const unsigned int plane_bytes = 3u * (unsigned int)H * (unsigned int)W;
      struct Row {
        unsigned long long m_fast;   // lanes finished by phase 1 in this row (their taps are in the window)
        unsigned long long m_gath;   // certified lanes inside the image whose taps were gathered from the planes instead
        float fu, fv;
        double Z;
        Cells t;
      };
      // window-relative floor (du, dv), fractions and z' of one pixel from its raw depth and its row's A, B, C
      auto coords = [](const unsigned int draw, const double Ar, const double Br, const double Cr, const double t0p,
                       const double t1p, const double t2, const double cxw, const double cyw, unsigned int& du,
                       unsigned int& dv, float& fu, float& fv, double& Zo) {
        const double z = (double)draw;
        const double X = fma(z, Ar, t0p), Y = fma(z, Br, t1p), Z = fma(z, Cr, t2);
        double rz = __builtin_amdgcn_rcp(Z);
        rz = fma(rz, fma(-Z, rz, 1.0), rz);
        const double uw = fma(X, rz, cxw), vw = fma(Y, rz, cyw);
        // the conversion saturates and maps NaN to 0: lanes with wild coordinates fail the window test
        int tu, tv;
        asm("v_cvt_i32_f64 %0, %1" : "=v"(tu) : "v"(uw));
        asm("v_cvt_i32_f64 %0, %1" : "=v"(tv) : "v"(vw));
        fu = (float)__builtin_amdgcn_fract(uw);
        fv = (float)__builtin_amdgcn_fract(vw);
        Zo = Z;
        du = (unsigned int)tu - 65537u; dv = (unsigned int)tv - 65537u;   // floor - window origin
      };
      auto project_row = [&](const int r) -> Row {
        Row o;
        const unsigned int draw = a_d[r];
        unsigned int du, dv;
        double Z;
        coords(draw, A, Bq, C, t0p, t1p, t2, cxw, cyw, du, dv, o.fu, o.fv, Z);
        o.Z = Z;
        A += kAr; Bq += kBr; C += kCr;
        unsigned long long m_valid0 = __builtin_amdgcn_ballot_w64(draw != 0u) & col_nz;
        if (!((rows_nz >> r) & 1u)) m_valid0 = 0ull;
        // certified: z' safely positive and both coordinates farther than the bound from every integer
        const unsigned long long m_cert = m_valid0 & __builtin_amdgcn_ballot_w64(Z > dz) &
                                          __builtin_amdgcn_ballot_w64(fmaxf(fabsf(o.fu - 0.5f), fabsf(o.fv - 0.5f)) <= half_eps);
        o.m_fast = m_cert & __builtin_amdgcn_ballot_w64(du < lim_u) & __builtin_amdgcn_ballot_w64(dv < lim_v);
        unsigned int idx = __umul24(dv, (unsigned int)WIN_PITCH) + du;
        asm("" : "+v"(idx));   // keeps the "* 8" of the cell size a shift: the reads are then known to be 8-byte aligned
        // (a select, not a masked read: reading under the lane mask saves the select and is 0.8 us slower per
        // launch for its two extra scalar instructions and the block boundary)
#ifdef SLAMHIP_AUDIT
        {
          const int row = row0 + r;
          double xe, ye, ze, x2e, y2e, z2e, ue, ve;
          backproject_px(cam, col, row, draw, xe, ye, ze);
          const bool v0e = (xe != 0.0) & (ye != 0.0) & (ze != 0.0);
          transform_px(p, xe, ye, ze, x2e, y2e, z2e);
          project_px(cam, x2e, y2e, z2e, ue, ve);
          const bool inside = ((rows_in >> r) & 1u) && col < W;
          const bool inb_e = v0e && in_window(ue, ve, (double)W, (double)H);
          const bool wv_e = v0e && in_window(ue, ve, (double)(W - 1), (double)(H - 1)) && (z2e > 0.0);
          const unsigned int iu_f = du + (unsigned int)wx0, iv_f = dv + (unsigned int)wy0;
          const bool inb_f = iu_f < (unsigned int)W && iv_f < (unsigned int)H;
          const bool wv_f = inb_f && iu_f + 1u < (unsigned int)W && iv_f + 1u < (unsigned int)H;
          const unsigned long long m_in = __builtin_amdgcn_ballot_w64(inside);
          audit_add(0, m_in);
          audit_add(1, m_cert & m_in);
          audit_add(2, m_in & m_valid0 & ~m_cert);
          audit_add(3, m_cert & m_in & __builtin_amdgcn_ballot_w64(inb_f != inb_e));
          audit_add(4, m_cert & m_in & __builtin_amdgcn_ballot_w64(wv_f != wv_e));
          audit_add(5, m_cert & m_in & __builtin_amdgcn_ballot_w64(
                           wv_e && ((int)iu_f != (int)floor(ue) || (int)iv_f != (int)floor(ve))));
          audit_add(6, m_in & (m_valid0 ^ __builtin_amdgcn_ballot_w64(v0e && inside)));
        }
#endif
        o.t = read_cells(win, __builtin_amdgcn_inverse_ballot_w64(o.m_fast) ? idx : 0u);
        o.m_gath = 0ull;
        const unsigned long long m_other = m_valid0 & ~o.m_fast;
        if (m_other != 0ull) {
          // Certified pixels outside the window: floor(u) = wx0 + du and floor(v) = wy0 + dv are the
          // reference's, u and v are no integers, so "0 <= u <= W" is "0 <= floor(u) <= W - 1" and
          // "u <= W - 1" is "floor(u) <= W - 2" (coordinates too large for int32 wrap or saturate to values
          // that stay outside).  A pixel outside the image is finished here; one inside it whose taps the
          // window does not hold gathers them below; every uncertified pixel goes to phase 2.
          const unsigned int iu = du + (unsigned int)wx0, iv = dv + (unsigned int)wy0;
          const unsigned long long m_out = m_cert & ~o.m_fast;
          const unsigned long long m_inb = m_out & __builtin_amdgcn_ballot_w64(iu < (unsigned int)W) &
                                           __builtin_amdgcn_ballot_w64(iv < (unsigned int)H);
          const unsigned long long m_wv = m_inb & __builtin_amdgcn_ballot_w64(iu + 1u < (unsigned int)W) &
                                          __builtin_amdgcn_ballot_w64(iv + 1u < (unsigned int)H);
          n_valid0 += (unsigned int)__popcll(m_out & ~m_wv);
          n_inb += (unsigned int)__popcll(m_inb & ~m_wv);
          const unsigned long long m_unc = m_valid0 & ~m_cert;
          if (m_unc != 0ull) rest_bits |= __builtin_amdgcn_inverse_ballot_w64(m_unc) ? (1u << r) : 0u;
          if (m_wv != 0ull) {
            // taps outside the window: the two tap pairs straight from the neighbour's planes, as loaded, into the
            // registers of the lane's (unused) window cells; they are put into the cells' form when the row is blended
            o.m_gath = m_wv;
#ifdef SLAMHIP_AUDIT
            audit_add(7, m_wv);
#endif
            if (__builtin_amdgcn_inverse_ballot_w64(m_wv)) {
              const unsigned int og = iv * (unsigned int)W + iu;
              const TapPair top = load_tap_pair(d2, c2, og, plane_bytes), bot = load_tap_pair(d2, c2, og + (unsigned int)W, plane_bytes);
              o.t.a.x = top.dd; o.t.a.y = bot.dd;
              o.t.b.x = top.qx; o.t.b.y = top.qy;
              o.t.c.x = bot.qx; o.t.c.y = bot.qy;
              o.t.e.x = top.sh | (bot.sh << 8);
            }
          }
        }
        n_fast += (unsigned int)__popcll(o.m_fast | o.m_gath);   // these pixels count for all three masks (added at the end)
        return o;
      };
      Row cur = project_row(0);
#pragma unroll
      for (int r = 0; r < RPW; ++r) {
        Row nxt = cur;
        if (r + 1 < RPW) nxt = project_row(r + 1);
        if (cur.m_gath != 0ull) {
          if (__builtin_amdgcn_inverse_ballot_w64(cur.m_gath)) {
            const TapPair top = {cur.t.a.x, cur.t.b.x, cur.t.b.y, cur.t.e.x & 0xffu};
            const TapPair bot = {cur.t.a.y, cur.t.c.x, cur.t.c.y, cur.t.e.x >> 8};
            tap_pair_cells(top, cur.t.a, cur.t.b);
            tap_pair_cells(bot, cur.t.c, cur.t.e);
          }
        }
        const unsigned long long m_blend = cur.m_fast | cur.m_gath;
        // (evaluated on every lane, whatever its cells hold: the mask and its count are scalar work)
        const unsigned long long m_geo = m_blend & __builtin_amdgcn_ballot_w64(taps_nonzero(cur.t));
        n_geo += (unsigned int)__popcll(m_geo);
        if (__builtin_amdgcn_inverse_ballot_w64(m_blend)) {
          const unsigned int a_rg = __builtin_amdgcn_perm(a_q[r].y, a_q[r].x, sel_rg);
          const unsigned int a_b = __builtin_amdgcn_perm(a_q[r].y, a_q[r].x, sel_b);
          blend_accumulate<UNIT>(acc, __builtin_amdgcn_inverse_ballot_w64(m_geo), cur.t, cur.fu, cur.fv, inv_factor_f,
                                 a_rg, a_b, cur.Z);
        }
        cur = nxt;
        __builtin_amdgcn_sched_barrier(0);   // no more than one row ahead: more only lengthens live ranges
      }
    } else {
      // no certificate (constants out of range, a degenerate pose) or no window: every pixel in the image
      rest_bits = __builtin_amdgcn_inverse_ballot_w64(col_in) ? rows_in : 0u;
#ifdef SLAMHIP_AUDIT
      for (int r = 0; r < RPW; ++r) {
        const unsigned long long m = __builtin_amdgcn_ballot_w64(((rest_bits >> r) & 1u) != 0u);
        audit_add(0, m);
        audit_add(2, m);
      }
#endif
    }

    SLAMHIP_STAMP(8);   // phase 1 done
    // ---- phase 2: the reference's exact operation order for the pixels phase 1 left ----
#ifdef SLAMHIP_ABL_LOADS_ONLY   // timing ablation only: every load and the staging, no projection, blend or sums
    {
      unsigned int chk = 0u;     // the anchors and one staged cell per lane stay alive in a store that never happens
#pragma unroll
      for (int r = 0; r < RPW; ++r) chk ^= a_d[r] ^ a_q[r].x ^ a_q[r].y;
      const uint2 c0 = win[(lane * 7 + wave) % (WIN_PITCH * WIN_ROWS)];
      chk ^= c0.x ^ c0.y;
      if (chk == 0x9e3779b9u) counts[0] = chk;
      rest_bits = 0u;
    }
#endif
#ifdef SLAMHIP_ABL_NO_PHASE2   // timing ablation only: results are wrong
    rest_bits = 0u;
#endif
    if (__builtin_amdgcn_ballot_w64(rest_bits != 0u) != 0ull) {
      const double dW = (double)W, dH = (double)H, dW1 = (double)(W - 1), dH1 = (double)(H - 1);
      // (single-neighbour kernel: camera, pose and plane pointers from the kernel arguments, see FusedKernArgs)
      Cam cam2 = cam;
      Pose p2 = p;
      const uint16_t* d1x = d1b;
      const uint8_t* c1x = c1b;
      const uint16_t* d2x = d2;
      const uint8_t* c2x = c2;
      if constexpr (K1) {
        const KernArgPtr q = late_kernargs();
        long long bq = b, bjq = bj;     // (opaque: the plane offsets are multiplied out here, not kept from the start)
        asm volatile("" : "+s"(bq), "+s"(bjq));
        cam2.fx = q->a.cam.fx; cam2.fy = q->a.cam.fy; cam2.cx = q->a.cam.cx; cam2.cy = q->a.cam.cy;
        cam2.factor = q->a.cam.factor;
        p2 = load_pose_const(q->rel + 16 * bjq);
        const long long HWq = (long long)q->a.H * q->a.W;
        d1x = q->depth1 + bq * HWq; c1x = q->rgb1 + 3 * bq * HWq;
        d2x = q->depth2 + bjq * HWq; c2x = q->rgb2 + 3 * bjq * HWq;
      }
#pragma unroll 1
      for (int r = 0; r < RPW; ++r) {
        const bool need = (rest_bits >> r) & 1u;
        if (__builtin_amdgcn_ballot_w64(need) == 0ull) continue;
        bool valid0 = false, inb = false, wv = false, geo = false;
        if (need) {
          const int row = row0 + r;
          const long long o = (long long)row * W + col;
          const unsigned int draw = d1x[o];
          double x, y, z, x2, y2, z2, u, v;
          backproject_px(cam2, col, row, draw, x, y, z);
          valid0 = (x != 0.0) & (y != 0.0) & (z != 0.0);
          transform_px(p2, x, y, z, x2, y2, z2);
          project_px(cam2, x2, y2, z2, u, v);
          inb = valid0 && in_window(u, v, dW, dH);
          wv = valid0 && in_window(u, v, dW1, dH1) && (z2 > 0.0);
          if (wv) {
            // 0 <= u <= W - 1 and 0 <= v <= H - 1: truncation is floor
            const double fl_u = floor(u), fl_v = floor(v);
            const int iu = (int)fl_u, iv = (int)fl_v;
            const float fu = (float)(u - fl_u), fv = (float)(v - fl_v);
            const unsigned int du = (unsigned int)(iu - wx0), dv = (unsigned int)(iv - wy0);
            Cells t;
            if (du < lim_u && dv < lim_v) {
              t = read_cells(win, dv * WIN_PITCH + du);
            } else {
              const int x1 = min(iu + 1, W - 1), y1 = min(iv + 1, H - 1);
              t.a = global_cell(d2x, c2x, W, iu, iv); t.b = global_cell(d2x, c2x, W, x1, iv);
              t.c = global_cell(d2x, c2x, W, iu, y1); t.e = global_cell(d2x, c2x, W, x1, y1);
            }
            const uint8_t* cp = c1x + 3 * o;
            const unsigned int rgbx = (unsigned int)cp[0] | ((unsigned int)cp[1] << 8) | ((unsigned int)cp[2] << 16);
            const unsigned int a_rg = __builtin_amdgcn_perm(0u, rgbx, 0x0c010c00u);
            const unsigned int a_b = __builtin_amdgcn_perm(0u, rgbx, 0x0c0c0c02u);
            geo = taps_nonzero(t);
            blend_accumulate<UNIT>(acc, geo, t, fu, fv, inv_factor_f, a_rg, a_b, z2);
          }
        }
        n_valid0 += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(valid0));
        n_inb += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(inb));
        n_warp += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(wv));
        n_geo += (unsigned int)__popcll(__builtin_amdgcn_ballot_w64(geo));
      }
    }

    SLAMHIP_STAMP(9);   // phase 2 done
    // ---- reduce: DPP within the wave, LDS across the four waves, one atomic each ----
    // (64 lanes adding to one LDS word serialise in the LDS pipe and stall the
    // wave far longer than the 6 DPP steps cost)
    const float t0 = wave_sum63(acc.ap) * COLOUR_UNSCALE, t1 = wave_sum63(acc.sp) * (COLOUR_UNSCALE * COLOUR_UNSCALE);
    const float t2s = wave_sum63((float)acc.ag), t3 = wave_sum63((float)acc.sg);
    if (lane == 63) {
      s_sum[wave][0] = t0; s_sum[wave][1] = t1; s_sum[wave][2] = t2s; s_sum[wave][3] = t3;
      s_cnt[wave][0] = n_valid0 + n_fast; s_cnt[wave][1] = n_inb + n_fast; s_cnt[wave][2] = n_warp + n_fast; s_cnt[wave][3] = n_geo;
    }
    __syncthreads();   // also: every gather is done before the next neighbour restages the window
    SLAMHIP_STAMP(10);  // all waves reduced
    if (wave == 0) {
      PairAcc* pa = ws + bj;
      double* sums_out = sums;
      unsigned int* counts_out = counts;
      unsigned int last_ticket = (unsigned int)tiles - 1u;
      long long bjo = bj;
      // (the lane number is made opaque here: the lane masks and output addresses are otherwise computed ahead of
      // the rows and, with every register taken there, parked in vector lanes or scratch memory)
      unsigned int pl = (unsigned int)lane;
      asm volatile("" : "+v"(pl));
      if constexpr (K1) {
        const KernArgPtr q = late_kernargs();
        asm volatile("" : "+s"(bjo));
        pa = q->ws + bjo;
        sums_out = q->sums;
        counts_out = q->counts;
        last_ticket = (unsigned int)(q->a.tiles_x * q->a.tiles_y) - 1u;
      }
      if (pl < 8) {
        const int q = (int)pl & 3;
        double t = (double)s_sum[0][q] + (double)s_sum[1][q];
        if (NW == 4) t += (double)s_sum[2][q] + (double)s_sum[3][q];
        unsigned int c = s_cnt[0][q] + s_cnt[1][q];
        if (NW == 4) c += s_cnt[2][q] + s_cnt[3][q];
        atomicAdd(&pa->vals[pl], pl < 4 ? t : (double)c);
      }
#ifdef SLAMHIP_ABL_NO_TICKET     // timing ablation only: nothing is published
      continue;
#endif
      // every add of this workgroup has been performed before its ticket is drawn
      __builtin_amdgcn_s_waitcnt(0);  // vmcnt(0) expcnt(0) lgkmcnt(0)
      unsigned int t = 0;
      if (pl == 0) t = atomicAdd(&pa->ticket, 1u);
      t = __builtin_amdgcn_readfirstlane(t);
      if (t == last_ticket) {
        // last workgroup of this pair: totals out, accumulators back to zero
        if (pl < 8) {
          const double v = atomicExch(&pa->vals[pl], 0.0);
          if (pl < 4) sums_out[bjo * SLAMHIP_NSUMS + pl] = v;
          else counts_out[bjo * SLAMHIP_NCOUNTS + (pl - 4)] = (unsigned int)v;
        } else if (pl == 8) {
          atomicExch(&pa->ticket, 0u);
        }
      }
      SLAMHIP_STAMP(11);  // ticket drawn
#ifdef SLAMHIP_STAMPS
      if (a.stamps && lane == 0) a.stamps[(size_t)wg_linear * 16 + 14] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    // the next neighbour's first barrier (after the window prediction) orders these reads of s_sum /
    // s_cnt before that neighbour's writes
  }
}

static bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

// Accumulator records are cleared with a kernel (zero_fill.h), never with a memset: callers may record these entry
// points in a HIP graph, where a memset becomes a node of its own kind; kernel nodes alone replay in order.
static hipError_t clear_records(void* workspace, size_t bytes, hipStream_t s) { return zero_async(workspace, bytes, s); }

}  // namespace slamhip

using namespace slamhip;

// Argument checks and launch shared by the two fused entry points; `workspace` holds at least
// B * k zero-filled PairAcc records.
static int launch_pair_loss(const char* who, const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                            const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor, int B,
                            int k, int H, int W, double* sums, uint32_t* counts, void* workspace, hipStream_t s) {
  SLAMHIP_REQUIRE(depth1 && rgb1 && depth2 && rgb2 && rel4x4 && K && sums && counts && workspace, SLAMHIP_E_ARG,
                  "%s: null pointer", who);
  SLAMHIP_REQUIRE(H <= 32767 && W <= 65535 && (long long)H * W <= 0x20000000LL, SLAMHIP_E_SIZE,
                  "%s: image larger than 65535 x 32767 or 2^29 pixels", who);
  SLAMHIP_REQUIRE(aligned_to(workspace, 8) && aligned_to(sums, 8) && aligned_to(counts, 4) && aligned_to(depth1, 2) &&
                  aligned_to(depth2, 2) && aligned_to(rel4x4, 8), SLAMHIP_E_ARG,
                  "%s: misaligned workspace, input or output", who);
  // One workgroup per tile of one (anchor, neighbour) pair, 128 x 24 or 128 x 16 pixels on four waves: that kernel is
  // straight-line code (no loop over neighbours), needs 99 / 86 vector registers and 39 / 30 KB of LDS, and four / five
  // workgroups share a compute unit.  The form that walks an anchor's k neighbours in one workgroup and reads the
  // anchor tile once (168 registers, 128 x 32 tiles, three workgroups per compute unit) is slower although it
  // reads less - the kernel is bound by vector instructions, and the k pairs of an anchor tile run back to
  // back on one XCD, so the repeated anchor reads come from its L2.  Measured per launch of 64 pairs:
  // k = 1: 76.5 us against 85.2 us; k = 4 (16 anchors): 71.6 us against 91.7 us.  Tuning builds (-DSLAMHIP_TUNING_TILES) hold the other
  // tile shapes (SLAMHIP_TILE = 16x16, 16x8, 24x12, 32x16, 12x6, 20x10, 8x4) and the shared-anchor form
  // (SLAMHIP_SHARED_ANCHORS=1).
#ifdef SLAMHIP_TUNING_TILES
  static const int tile_env = []() {
    const char* e = getenv("SLAMHIP_TILE");
    const std::string v = e ? e : "";
    return v == "16x16" ? 1 : v == "16x8" ? 2 : v == "24x12" ? 3 : v == "32x16" ? 5 : v == "12x6" ? 6 :
           v == "20x10" ? 7 : v == "8x4" ? 8 : 0;
  }();
  static const bool shared_env = []() { const char* e = getenv("SLAMHIP_SHARED_ANCHORS"); return e && atoi(e) != 0; }();
  const bool shared_anchors = shared_env && k > 1;
#else
  constexpr int tile_env = 0;
  constexpr bool shared_anchors = false;
#endif
  // 128 x 16 tiles (five workgroups per compute unit) are 3-4 % ahead of 128 x 24 when neighbours share anchors
  // through L2.  For k = 1 the launch size decides: the smaller tile's workgroups live shorter, so the launch
  // fills and drains faster, the larger tile spends fewer instructions per pixel; measured at 640 x 480 per
  // launch of 16 / 32 / 64 / 96 / 128 / 256 pairs, 128 x 16 against 128 x 24: 20.0 / 21.3, 35.8 / 36.9, 64.1 / 65.2,
  // 94.8 / 93.5, 125.3 / 121.5, 246.2 / 238.9 us.  The crossover is near eight rounds of the chip's 1024 workgroup slots.
  const long long wgs24 = (long long)B * ((W + TILE_W - 1) / TILE_W) * ((H + 23) / 24);
  const int tile_cfg = tile_env ? tile_env : (shared_anchors ? 5 : (k == 1 && wgs24 > 8 * 1024) ? 3 : 2);
  const int tile_h = tile_cfg == 1 || tile_cfg == 2 ? 16 : tile_cfg == 3 ? 24 : tile_cfg == 6 ? 12 : tile_cfg == 7 ? 20 :
                     tile_cfg == 8 ? 8 : 32;
  FusedArgs a;
  PairAcc* ws = (PairAcc*)workspace;
  a.cam = Cam{K[0], K[1], K[2], K[3], factor};
  a.B = B; a.k = k; a.H = H; a.W = W;
  a.tiles_x = (W + TILE_W - 1) / TILE_W;
  a.tiles_y = (H + tile_h - 1) / tile_h;
  const long long groups = ((long long)B + 7) / 8;
  const long long grid_y = (long long)a.tiles_y * (shared_anchors ? 1 : k);
  SLAMHIP_REQUIRE(groups <= 65535 && grid_y <= 65535 && 8LL * a.tiles_x <= 0x7fffffffLL, SLAMHIP_E_SIZE,
                  "%s: grid too large (at most 524280 anchors and 65535 (tile row, neighbour) pairs per call)", who);
  a.inv_k = k > 1 ? (unsigned int)(0x100000000ULL / (unsigned long long)k) + 1u : 0u;
  // the window staging uses 16-byte / 8-byte vector loads: needs W % 8 == 0 and aligned planes
  a.stage_vec = (W % CHUNK == 0) && aligned_to(depth2, 16) && aligned_to(rgb2, 8);
  a.anchor_vec = (W % 4 == 0) && aligned_to(rgb1, 4);
  // phase 1 assumes that no intermediate comes near the denormal or overflow range ("x, y, z all non-zero"
  // is then "raw depth, col - cx and row - cy all non-zero"): constants in [2^-64, 2^64], principal point
  // in [2^-32, 2^32]; its error bound covers the roundings of the quotients below
  auto mid = [](double v) { const double m = fabs(v); return m >= 0x1p-64 && m <= 0x1p64; };
  auto pp = [](double v) { const double m = fabs(v); return m >= 0x1p-32 && m <= 0x1p32; };
#ifdef SLAMHIP_STAMPS
  a.stamps = g_stamps;
#else
  a.stamps = nullptr;
#endif
  a.cert_ok = mid(K[0]) && mid(K[1]) && mid(factor) && pp(K[2]) && pp(K[3]);
  a.fx_fy = K[0] / K[1]; a.fy_fx = K[1] / K[0];
  a.inv_fac = 1.0 / factor; a.inv_fx_fac = 1.0 / (K[0] * factor); a.inv_fy_fac = 1.0 / (K[1] * factor);
  a.axm = (float)(fmax(fabs(K[2]), fabs((double)(W - 1) - K[2])) / fabs(K[0]) * 1.0001);
  a.aym = (float)(fmax(fabs(K[3]), fabs((double)(H - 1) - K[3])) / fabs(K[1]) * 1.0001);
  a.extent = (float)((double)W + (double)H + fabs(K[2]) + fabs(K[3]) + 8.0);
  auto int_index = [](double c, int n) { return (c >= 0.0 && c < (double)n && c == floor(c)) ? (int)c : -1; };
  a.cx_col = int_index(K[2], W);
  a.cy_row = int_index(K[3], H);
  const bool unit = factor == 1.0;
  const dim3 grid(8u * (unsigned)a.tiles_x, (unsigned)grid_y, (unsigned)groups);
#define SLAMHIP_LAUNCH_FUSED(TH_, RPW_, K1_)                                                                         \
  do {                                                                                                             \
    const dim3 block(128 * (TH_ / RPW_));                                                                          \
    if (unit) hipLaunchKernelGGL((pair_loss_fused_kernel<TH_, RPW_, true, K1_>), grid, block, 0, s, a, depth1,     \
                                 rgb1, depth2, rgb2, rel4x4, ws, sums, counts);                                    \
    else hipLaunchKernelGGL((pair_loss_fused_kernel<TH_, RPW_, false, K1_>), grid, block, 0, s, a, depth1, rgb1,   \
                            depth2, rgb2, rel4x4, ws, sums, counts);                                               \
  } while (0)
#ifdef SLAMHIP_TUNING_TILES
  if (shared_anchors) {
    switch (tile_cfg) {
      case 3: SLAMHIP_LAUNCH_FUSED(24, 12, false); break;
      default: SLAMHIP_LAUNCH_FUSED(32, 16, false); break;
    }
  } else {
    switch (tile_cfg) {
      case 1: SLAMHIP_LAUNCH_FUSED(16, 16, true); break;
      case 2: SLAMHIP_LAUNCH_FUSED(16, 8, true); break;
      case 5: SLAMHIP_LAUNCH_FUSED(32, 16, true); break;
      case 6: SLAMHIP_LAUNCH_FUSED(12, 6, true); break;
      case 7: SLAMHIP_LAUNCH_FUSED(20, 10, true); break;
      case 8: SLAMHIP_LAUNCH_FUSED(8, 4, true); break;
      default: SLAMHIP_LAUNCH_FUSED(24, 12, true); break;
    }
  }
#else
  if (tile_cfg == 3) SLAMHIP_LAUNCH_FUSED(24, 12, true);
  else SLAMHIP_LAUNCH_FUSED(16, 8, true);
#endif
#undef SLAMHIP_LAUNCH_FUSED
  SLAMHIP_HIP(hipGetLastError());
  return SLAMHIP_OK;
}

extern "C" {

#ifdef SLAMHIP_STAMPS
// diagnostic builds only: 16 x uint64 per workgroup (device memory); NULL switches the stamps off
void slamhip_debug_set_stamps(unsigned long long* p) { g_stamps = p; }
#endif

size_t slamhip_pair_loss_workspace_bytes(int B, int k) {
  if (B <= 0 || k <= 0) return 0;
  return sizeof(PairAcc) * (size_t)B * (size_t)k;
}

int slamhip_pair_loss_fused(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                            const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor,
                            int B, int k, int H, int W, double* sums, uint32_t* counts, void* workspace,
                            size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && k > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "pair_loss_fused: bad shape B=%d k=%d H=%d W=%d", B, k, H, W);
  if (B == 0) return SLAMHIP_OK;  // an empty batch has no buffers to check
  SLAMHIP_REQUIRE(workspace != nullptr, SLAMHIP_E_ARG, "pair_loss_fused: null pointer");
  SLAMHIP_REQUIRE(workspace_bytes >= slamhip_pair_loss_workspace_bytes(B, k), SLAMHIP_E_SIZE,
                  "pair_loss_fused: workspace of %zu bytes, need %zu", workspace_bytes,
                  slamhip_pair_loss_workspace_bytes(B, k));
  return launch_pair_loss("pair_loss_fused", depth1, rgb1, depth2, rgb2, rel4x4, K, factor, B, k, H, W, sums,
                          counts, workspace, (hipStream_t)stream);
}

int slamhip_pair_loss_reset_workspace(void* workspace, size_t workspace_bytes, void* stream) {
  SLAMHIP_REQUIRE(workspace || workspace_bytes == 0, SLAMHIP_E_ARG, "pair_loss_reset_workspace: null workspace");
  if (workspace_bytes == 0) return SLAMHIP_OK;
  SLAMHIP_HIP(clear_records(workspace, workspace_bytes, (hipStream_t)stream));
  return SLAMHIP_OK;
}

// ---- the fused kernel for callers that cannot allocate ---------------------
// The accumulator records live in a file-scope device array instead of a caller's
// workspace.  Every GPU holds its own copy of the array at its own address: the
// address is looked up once per device (with that device current) and only that
// pointer, never the array's name, is used by host code afterwards.
static __device__ __align__(256) unsigned char scratch[SLAMHIP_STATIC_MAX_PAIRS * sizeof(PairAcc)];
constexpr int MAX_DEVICES = 64;
static std::atomic<void*> scratch_ptr[MAX_DEVICES];

int slamhip_pair_loss_fused_static(const uint16_t* depth1, const uint8_t* rgb1, const uint16_t* depth2,
                                   const uint8_t* rgb2, const double* rel4x4, const double K[4], double factor,
                                   int B, int k, int H, int W, double* sums, uint32_t* counts, void* stream) {
  SLAMHIP_REQUIRE(B >= 0 && k > 0 && H > 0 && W > 0, SLAMHIP_E_ARG,
                  "pair_loss_fused_static: bad shape B=%d k=%d H=%d W=%d", B, k, H, W);
  if (B == 0) return SLAMHIP_OK;
  SLAMHIP_REQUIRE((long long)B * k <= SLAMHIP_STATIC_MAX_PAIRS, SLAMHIP_E_SIZE,
                  "pair_loss_fused_static: %lld (anchor, neighbour) pairs, the built-in scratch holds %d "
                  "(use slamhip_pair_loss_fused with a workspace)", (long long)B * k, SLAMHIP_STATIC_MAX_PAIRS);
  int dev = -1;
  SLAMHIP_HIP(hipGetDevice(&dev));
  SLAMHIP_REQUIRE(dev >= 0 && dev < MAX_DEVICES, SLAMHIP_E_SIZE, "pair_loss_fused_static: device %d", dev);
  hipStream_t s = (hipStream_t)stream;
  if (s) {
    // every GPU holds its own copy of the array at its own address: the address looked up for the current device
    // is good only for a stream of that device
    int sdev = -1;
    SLAMHIP_HIP(hipStreamGetDevice(s, &sdev));
    SLAMHIP_REQUIRE(sdev == dev, SLAMHIP_E_ARG,
                    "pair_loss_fused_static: the stream belongs to device %d, the current device is %d (make the "
                    "stream's device current before the call)", sdev, dev);
  }
  void* ptr = scratch_ptr[dev].load(std::memory_order_acquire);
  if (!ptr) {   // first call on this device (racing first calls look up the same address)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    if (s) SLAMHIP_HIP(hipStreamIsCapturing(s, &cap));
    SLAMHIP_REQUIRE(cap == hipStreamCaptureStatusNone, SLAMHIP_E_ARG,
                    "pair_loss_fused_static: first call on device %d inside a stream capture (the address lookup "
                    "of the scratch array is no stream operation: call once eagerly before capturing)", dev);
    SLAMHIP_HIP(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    SLAMHIP_REQUIRE(ptr != nullptr, SLAMHIP_E_HIP, "pair_loss_fused_static: no address for the scratch array");
    scratch_ptr[dev].store(ptr, std::memory_order_release);
  }
  // the array keeps whatever the last call (possibly an interrupted one) left: clear the records of this call
  SLAMHIP_HIP(clear_records(ptr, sizeof(PairAcc) * (size_t)B * (size_t)k, s));
  return launch_pair_loss("pair_loss_fused_static", depth1, rgb1, depth2, rgb2, rel4x4, K, factor, B, k, H, W,
                          sums, counts, ptr, s);
}

#ifdef SLAMHIP_AUDIT
// audit build only: copies the eight counters to the host (after draining the device) and optionally clears them
int slamhip_audit_read(unsigned long long out[8], int reset) {
  SLAMHIP_HIP(hipDeviceSynchronize());
  SLAMHIP_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_audit), 8 * sizeof(unsigned long long)));
  if (reset) {
    const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    SLAMHIP_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_audit), z, sizeof(z)));
  }
  return SLAMHIP_OK;
}
#endif

}  // extern "C"
