/*
 * CPU oracle in plain C for the RGB-D frame-pair path.
 *
 * TEST INFRASTRUCTURE ONLY: the same restatement as oracle/slam_oracle.py
 * (which is pinned against golden vectors of the reference), in C so that
 * bench.py can time a CPU baseline on the GPU box's host cores and tests can
 * check large frames quickly.  Nothing that ships links or loads this file.
 *
 * Reference followed: slam/preprocess/test_import.py:148-176 (geometry, in its
 * float64 operation order; np.dot = dgemm = fused accumulate over k), plus the
 * dense extension (pixel index, bilinear warp, residual sums) defined in
 * oracle/slam_oracle.py.  Build with -ffp-contract=off (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define M_VALID0 1
#define M_INB 2
#define M_WARP 4
#define M_GEO 8

typedef struct {
  double x, y, z, x2, y2, z2, u, v;
  int valid0, inb, wv;
} proj_t;

static inline void project(const double* K, double factor, const double* T, int uu, int vv,
                           uint16_t raw, int H, int W, proj_t* p) {
  const double fx = K[0], fy = K[1], cx = K[2], cy = K[3];
  /* test_import.py:153-155 */
  p->z = (double)raw / factor;
  p->x = (((double)uu - cx) * p->z) / fx;
  p->y = (((double)vv - cy) * p->z) / fy;
  p->valid0 = (p->x != 0.0) && (p->y != 0.0) && (p->z != 0.0);
  /* test_import.py:160 */
  p->x2 = fma(p->z, T[8], fma(p->y, T[4], p->x * T[0])) + T[3];
  p->y2 = fma(p->z, T[9], fma(p->y, T[5], p->x * T[1])) + T[7];
  p->z2 = fma(p->z, T[10], fma(p->y, T[6], p->x * T[2])) + T[11];
  /* test_import.py:165-167 */
  p->u = (p->x2 * fx) / p->z2 + cx;
  p->v = (p->y2 * fy) / p->z2 + cy;
  /* test_import.py:174 with dim = [W, H] */
  p->inb = p->valid0 && (0.0 <= p->u) && (p->u <= (double)W) && (0.0 <= p->v) && (p->v <= (double)H);
  p->wv = p->valid0 && (0.0 <= p->u) && (p->u <= (double)(W - 1)) && (0.0 <= p->v) &&
          (p->v <= (double)(H - 1)) && (p->z2 > 0.0);
}

static inline int32_t pixel_index(double c) {
  if (c != c) return INT32_MIN;
  double f = floor(c);
  if (f <= -2147483648.0) return INT32_MIN;
  if (f >= 2147483647.0) return INT32_MAX;
  return (int32_t)f;
}

static inline float lerp2(float a, float b, float c, float e, float fu, float fv) {
  const float top = a + fu * (b - a);
  const float bot = c + fu * (e - c);
  return top + fv * (bot - top);
}

/* One anchor/neighbour pair.  Any of the per-pixel outputs may be NULL. */
__attribute__((target_clones("default", "fma")))
void slam_oracle_pair(const uint16_t* d1, const uint8_t* c1, const uint16_t* d2, const uint8_t* c2,
                      const double* T, const double* K, double factor, int H, int W, double* sums,
                      uint32_t* counts, float* warped_rgb, float* warped_depth, float* r_photo,
                      float* r_geo, uint8_t* mask, int32_t* iu_iv, double* uvd, double* xyz,
                      double* xyz2) {
  double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
  uint32_t n0 = 0, n1 = 0, n2 = 0, n3 = 0;
  const float factor_f = (float)factor;
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      const long i = (long)v * W + u;
      proj_t p;
      project(K, factor, T, u, v, d1[i], H, W, &p);
      float w[4] = {0, 0, 0, 0}, rp[3] = {0, 0, 0}, rg = 0;
      int geo = 0;
      if (p.wv) {
        const double flu = floor(p.u), flv = floor(p.v);
        const int x0 = (int)flu, y0 = (int)flv;
        const int x1 = x0 + 1 < W - 1 ? x0 + 1 : W - 1, y1 = y0 + 1 < H - 1 ? y0 + 1 : H - 1;
        const float fu = (float)(p.u - flu), fv = (float)(p.v - flv);
        const long o00 = (long)y0 * W + x0, o01 = (long)y0 * W + x1, o10 = (long)y1 * W + x0,
                   o11 = (long)y1 * W + x1;
        for (int c = 0; c < 3; ++c)
          w[c] = lerp2((float)c2[3 * o00 + c], (float)c2[3 * o01 + c], (float)c2[3 * o10 + c],
                       (float)c2[3 * o11 + c], fu, fv);
        w[3] = lerp2((float)d2[o00], (float)d2[o01], (float)d2[o10], (float)d2[o11], fu, fv) / factor_f;
        geo = d2[o00] != 0 && d2[o01] != 0 && d2[o10] != 0 && d2[o11] != 0;
        for (int c = 0; c < 3; ++c) {
          rp[c] = w[c] - (float)c1[3 * i + c];
          s0 += fabs((double)rp[c]);
          s1 += (double)rp[c] * (double)rp[c];
        }
        if (geo) {
          rg = (float)((double)w[3] - p.z2);
          s2 += fabs((double)rg);
          s3 += (double)rg * (double)rg;
        }
      }
      n0 += p.valid0; n1 += p.inb; n2 += p.wv; n3 += geo;
      if (warped_rgb) { warped_rgb[3 * i] = w[0]; warped_rgb[3 * i + 1] = w[1]; warped_rgb[3 * i + 2] = w[2]; }
      if (warped_depth) warped_depth[i] = w[3];
      if (r_photo) { r_photo[3 * i] = rp[0]; r_photo[3 * i + 1] = rp[1]; r_photo[3 * i + 2] = rp[2]; }
      if (r_geo) r_geo[i] = rg;
      if (mask) mask[i] = (uint8_t)((p.valid0 ? M_VALID0 : 0) | (p.inb ? M_INB : 0) | (p.wv ? M_WARP : 0) | (geo ? M_GEO : 0));
      if (iu_iv) { iu_iv[2 * i] = pixel_index(p.u); iu_iv[2 * i + 1] = pixel_index(p.v); }
      if (uvd) { uvd[3 * i] = p.u; uvd[3 * i + 1] = p.v; uvd[3 * i + 2] = p.z2; }
      if (xyz) { xyz[3 * i] = p.x; xyz[3 * i + 1] = p.y; xyz[3 * i + 2] = p.z; }
      if (xyz2) { xyz2[3 * i] = p.x2; xyz2[3 * i + 1] = p.y2; xyz2[3 * i + 2] = p.z2; }
    }
  if (sums) { sums[0] = s0; sums[1] = s1; sums[2] = s2; sums[3] = s3; }
  if (counts) { counts[0] = n0; counts[1] = n1; counts[2] = n2; counts[3] = n3; }
}

void slam_oracle_pair_loss_mt(const uint16_t* d1, const uint8_t* c1, const uint16_t* d2, const uint8_t* c2,
                              const double* rel, const double* K, double factor, int B, int k, int H, int W,
                              double* sums, uint32_t* counts, int threads);

/* Batched front end of the reduction: depth1 [B,H,W], rgb1 [B,H,W,3], depth2
 * [B,k,H,W], rgb2 [B,k,H,W,3], rel [B,k,16] -> sums [B,k,4], counts [B,k,4]. */
void slam_oracle_pair_loss(const uint16_t* d1, const uint8_t* c1, const uint16_t* d2, const uint8_t* c2,
                           const double* rel, const double* K, double factor, int B, int k, int H, int W,
                           double* sums, uint32_t* counts) {
  slam_oracle_pair_loss_mt(d1, c1, d2, c2, rel, K, factor, B, k, H, W, sums, counts, 1);
}

/* The same over `threads` host threads (OpenMP over the independent pairs): the
 * all-cores CPU baseline of the benchmark.  Results do not depend on `threads`. */
void slam_oracle_pair_loss_mt(const uint16_t* d1, const uint8_t* c1, const uint16_t* d2, const uint8_t* c2,
                              const double* rel, const double* K, double factor, int B, int k, int H, int W,
                              double* sums, uint32_t* counts, int threads) {
  const long HW = (long)H * W;
  const long n = (long)B * k;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads > 0 ? threads : 1)
  for (long bj = 0; bj < n; ++bj) {
    const long b = bj / k;
    slam_oracle_pair(d1 + b * HW, c1 + 3 * b * HW, d2 + bj * HW, c2 + 3 * bj * HW, rel + 16 * bj, K,
                     factor, H, W, sums + 4 * bj, counts + 4 * bj, 0, 0, 0, 0, 0, 0, 0, 0, 0);
  }
}
