// Host pose / label maths of the frame-pair path (float64), restating
// slam/preprocess/test_import.py:18-110 and the clamp / no-guard variants of
// slam/network/model_input.py:46-106 with their quirks kept.  Built with
// -ffp-contract=off: every product and sum below rounds once, fma() is written
// where the reference's BLAS call fuses.
#include <cmath>
#include <cstring>

#include "common.h"

namespace slamhip {

static thread_local char g_err[512];
char* error_buffer() { return g_err; }

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// test_import.py:18-47; the products run in float32 because the callers hand in
// float32 rows (test_import.py:199), the results are stored as float64.
static void quat_to_T(const float* g, double* T) {
  const float qx = g[3], qy = g[4], qz = g[5], qw = g[6];
  const float two = 2.0f;
  T[0] = (double)(((qw * qw + qx * qx) - qy * qy) - qz * qz);
  T[4] = (double)(two * (qx * qy + qw * qz));
  T[8] = (double)(two * (qz * qx - qw * qy));
  T[12] = 0.0;
  T[1] = (double)(two * (qx * qy - qw * qz));
  T[5] = (double)(((qw * qw - qx * qx) + qy * qy) - qz * qz);
  T[9] = (double)(two * (qy * qz + qw * qx));
  T[13] = 0.0;
  T[2] = (double)(two * (qz * qx + qw * qy));
  T[6] = (double)(two * (qy * qz - qw * qx));
  T[10] = (double)(((qw * qw - qx * qx) - qy * qy) + qz * qz);
  T[14] = 0.0;
  T[3] = (double)g[0];
  T[7] = (double)g[1];
  T[11] = (double)g[2];
  T[15] = 1.0;
}

// test_import.py:80-86
static void inverse_T(const double* T, double* inv) {
  std::memset(inv, 0, 16 * sizeof(double));
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) inv[i * 4 + j] = T[j * 4 + i];
  for (int i = 0; i < 3; ++i) {
    double acc = (-inv[i * 4 + 0]) * T[3];
    acc = std::fma(-inv[i * 4 + 1], T[7], acc);
    acc = std::fma(-inv[i * 4 + 2], T[11], acc);
    inv[i * 4 + 3] = acc;
  }
  inv[15] = 1.0;
}

// 4x4 product in the dgemm evaluation order (k ascending, fused accumulate).
static void matmul4(const double* A, const double* B, double* C) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double acc = A[i * 4 + 0] * B[0 * 4 + j];
      for (int k = 1; k < 4; ++k) acc = std::fma(A[i * 4 + k], B[k * 4 + j], acc);
      C[i * 4 + j] = acc;
    }
}

// 3x3 inverse by Gauss-Jordan elimination with partial pivoting; a zero pivot
// leaves NaN (the reference's LinAlgError / NaN cases are not relied upon).
static void inverse3(const double* M, double* out) {
  double a[3][6];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      a[i][j] = M[i * 3 + j];
      a[i][3 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 3; ++c) {
    int p = c;
    for (int r = c + 1; r < 3; ++r)
      if (std::fabs(a[r][c]) > std::fabs(a[p][c])) p = r;
    if (p != c)
      for (int j = 0; j < 6; ++j) {
        double t = a[c][j];
        a[c][j] = a[p][j];
        a[p][j] = t;
      }
    const double piv = a[c][c];
    for (int j = 0; j < 6; ++j) a[c][j] /= piv;
    for (int r = 0; r < 3; ++r) {
      if (r == c) continue;
      const double f = a[r][c];
      for (int j = 0; j < 6; ++j) a[r][j] -= f * a[c][j];
    }
  }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) out[i * 3 + j] = a[i][3 + j];
}

// test_import.py:50-77, model_input.py:46-76 (clamp)
static void T_to_twist(const double* T, int clamp, double* tw) {
  for (int i = 0; i < 6; ++i) tw[i] = 0.0;
  double arg = (((T[0] + T[5]) + T[10]) - 1.0) / 2.0;
  if (arg > 1.0) return;
  if (clamp && std::fabs(arg) > 1.0) arg = 1.0 * arg / std::fabs(arg);
  const double th = std::acos(arg);
  const double s = 1.0 / (2.0 * std::sin(th));
  double w[3];
  w[0] = s * (T[9] - T[6]) * th;
  w[1] = s * (T[2] - T[8]) * th;
  w[2] = s * (T[4] - T[1]) * th;
  const double ww = (w[0] * w[0] + w[1] * w[1]) + w[2] * w[2];
  const double w_abs = std::sqrt(ww);
  if (w_abs == 0.0) return;
  const double what[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  double om[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double imr = ((i == k) ? 1.0 : 0.0) - T[i * 4 + k];
        acc = (k == 0) ? imr * what[k * 3 + j] : std::fma(imr, what[k * 3 + j], acc);
      }
      // `w * w^T` is a 1x1 matrix in the reference: |w|^2 lands on all nine entries
      om[i * 3 + j] = (acc + ww) / (w_abs * w_abs);
    }
  double inv[9];
  inverse3(om, inv);
  for (int i = 0; i < 3; ++i) {
    double acc = inv[i * 3 + 0] * T[3];
    acc = std::fma(inv[i * 3 + 1], T[7], acc);
    acc = std::fma(inv[i * 3 + 2], T[11], acc);
    tw[i] = acc;
  }
  tw[3] = w[0];
  tw[4] = w[1];
  tw[5] = w[2];
}

// test_import.py:89-110, model_input.py:88-106 (no guard)
static void twist_to_T(const double* tw, int guard, double* T) {
  const double* v = tw;
  const double* w = tw + 3;
  const double w_abs = std::sqrt((w[0] * w[0] + w[1] * w[1]) + w[2] * w[2]);
  std::memset(T, 0, 16 * sizeof(double));
  T[15] = 1.0;
  if (w_abs == 0.0) {
    T[0] = T[5] = T[10] = 1.0;
    if (!guard) T[3] = T[7] = T[11] = NAN;  // rotation I, translation 0/0 (see oracle)
    return;
  }
  const double what[9] = {0.0, -w[2], w[1], w[2], 0.0, -w[0], -w[1], w[0], 0.0};
  double what2[9], rot[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = what[i * 3 + 0] * what[0 * 3 + j];
      acc = std::fma(what[i * 3 + 1], what[1 * 3 + j], acc);
      acc = std::fma(what[i * 3 + 2], what[2 * 3 + j], acc);
      what2[i * 3 + j] = acc;
    }
  const double sn = std::sin(w_abs), cs = std::cos(w_abs);
  for (int i = 0; i < 9; ++i) {
    const double eye = (i % 4 == 0) ? 1.0 : 0.0;
    rot[i] = (eye + what[i] / w_abs * sn) + what2[i] / (w_abs * w_abs) * (1.0 - cs);
  }
  // t = ((I - rot) . w_hat . v + (w^T w) . v) / |w|^2   (w^T w is the 3x3 outer product)
  double m[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) {
      double acc = 0.0;
      for (int k = 0; k < 3; ++k) {
        const double imr = ((i == k) ? 1.0 : 0.0) - rot[i * 3 + k];
        acc = (k == 0) ? imr * what[k * 3 + j] : std::fma(imr, what[k * 3 + j], acc);
      }
      m[i * 3 + j] = acc;
    }
  for (int i = 0; i < 3; ++i) {
    double a = m[i * 3 + 0] * v[0];
    a = std::fma(m[i * 3 + 1], v[1], a);
    a = std::fma(m[i * 3 + 2], v[2], a);
    double b = (w[i] * w[0]) * v[0];
    b = std::fma(w[i] * w[1], v[1], b);
    b = std::fma(w[i] * w[2], v[2], b);
    T[i * 4 + 3] = (a + b) / (w_abs * w_abs);
    for (int j = 0; j < 3; ++j) T[i * 4 + j] = rot[i * 3 + j];
  }
}

// slam/preprocess/import_data.py:12-42 - position + rotation vector of an absolute pose,
// the rotation rebuilt from Euler angles of the quaternion READ AS (q0..q3) = (qx, qy, qz, qw)
// in the formulas of a (w, x, y, z) quaternion (sic), operations in the reference's order.
static void absolute_position(const double* g, double* out) {
  const double q0 = g[3], q1 = g[4], q2 = g[5], q3 = g[6];
  const double phi = atan2(2 * (q0 * q1 + q2 * q3), 1 - 2 * (q1 * q1 + q2 * q2));
  const double theta = asin(2 * (q0 * q2 - q3 * q1));
  const double psi = atan2(2 * (q0 * q3 + q1 * q2), 1 - 2 * (q2 * q2 + q3 * q3));
  const double ct = cos(theta), st = sin(theta), cp = cos(phi), sp = sin(phi), cs = cos(psi), ss = sin(psi);
  double R[3][3];
  R[0][0] = ct * cp;
  R[1][0] = ct * sp;
  R[2][0] = -st;
  R[0][1] = ss * st * cp - cs * sp;
  R[1][1] = ss * st * sp + cs * cp;
  R[2][1] = ss * ct;
  R[0][2] = cs * st * cp + ss * sp;
  R[1][2] = cs * st * sp - ss * cp;
  R[2][2] = cs * ct;
  const double w_abs = acos(((R[0][0] + R[1][1] + R[2][2]) - 1) / 2);
  const double f = 1 / (2 * sin(w_abs));
  out[0] = g[0]; out[1] = g[1]; out[2] = g[2];
  out[3] = f * (R[2][1] - R[1][2]) * w_abs;
  out[4] = f * (R[0][2] - R[2][0]) * w_abs;
  out[5] = f * (R[1][0] - R[0][1]) * w_abs;
}

}  // namespace slamhip

using namespace slamhip;

extern "C" {

const char* slamhip_last_error(void) { return error_buffer(); }
int slamhip_version(void) { return 100; }

int slamhip_quat_to_transformation(const float* gt7, int n, double* T4x4) {
  SLAMHIP_REQUIRE(gt7 && T4x4 && n >= 0, SLAMHIP_E_ARG, "quat_to_transformation: null pointer or n<0");
  for (int i = 0; i < n; ++i) quat_to_T(gt7 + 7 * i, T4x4 + 16 * i);
  return SLAMHIP_OK;
}

int slamhip_inverse_trans(const double* T4x4, int n, double* inv4x4) {
  SLAMHIP_REQUIRE(T4x4 && inv4x4 && n >= 0, SLAMHIP_E_ARG, "inverse_trans: null pointer or n<0");
  for (int i = 0; i < n; ++i) inverse_T(T4x4 + 16 * i, inv4x4 + 16 * i);
  return SLAMHIP_OK;
}

int slamhip_trans_to_twist(const double* T4x4, int n, int clamp, double* twist6) {
  SLAMHIP_REQUIRE(T4x4 && twist6 && n >= 0, SLAMHIP_E_ARG, "trans_to_twist: null pointer or n<0");
  for (int i = 0; i < n; ++i) T_to_twist(T4x4 + 16 * i, clamp, twist6 + 6 * i);
  return SLAMHIP_OK;
}

int slamhip_twist_to_trans(const double* twist6, int n, int guard, double* T4x4) {
  SLAMHIP_REQUIRE(twist6 && T4x4 && n >= 0, SLAMHIP_E_ARG, "twist_to_trans: null pointer or n<0");
  for (int i = 0; i < n; ++i) twist_to_T(twist6 + 6 * i, guard, T4x4 + 16 * i);
  return SLAMHIP_OK;
}

int slamhip_pose_batch(const float* gt7_a, const float* gt7_b, int n, int clamp, double* rel4x4,
                       double* twist6) {
  SLAMHIP_REQUIRE(gt7_a && gt7_b && n >= 0, SLAMHIP_E_ARG, "pose_batch: null pointer or n<0");
  for (int i = 0; i < n; ++i) {
    double Ta[16], Tb[16], inv[16], rel[16];
    quat_to_T(gt7_a + 7 * i, Ta);
    quat_to_T(gt7_b + 7 * i, Tb);
    inverse_T(Ta, inv);
    matmul4(inv, Tb, rel);
    if (rel4x4) std::memcpy(rel4x4 + 16 * i, rel, sizeof(rel));
    if (twist6) T_to_twist(rel, clamp, twist6 + 6 * i);
  }
  return SLAMHIP_OK;
}

int slamhip_absolute_position(const double* gt7, int n, double* pos6) {
  SLAMHIP_REQUIRE(n >= 0 && (n == 0 || (gt7 && pos6)), SLAMHIP_E_ARG, "absolute_position: bad argument");
  for (int i = 0; i < n; ++i) absolute_position(gt7 + 7 * i, pos6 + 6 * i);
  return SLAMHIP_OK;
}

}  // extern "C"
