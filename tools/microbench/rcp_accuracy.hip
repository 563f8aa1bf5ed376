// Accuracy on gfx950 of the reciprocal seeds a fast float64 division can start from:
// raw v_rcp_f64, v_rcp_f32 of the rounded operand, and each after one Newton step in float64.
// Prints the largest relative error (in units of 2^-k) over random operands of many magnitudes.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(const double* z, int n, double* err) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double b = z[i];
  const double exact = 1.0 / b;
  const double r64 = __builtin_amdgcn_rcp(b);
  const double r32 = (double)__builtin_amdgcn_rcpf((float)b);
  auto newton = [&](double r) { const double e = fma(-b, r, 1.0); return fma(r, e, r); };
  auto rel = [&](double r) { return fabs(r - exact) / fabs(exact); };
  err[4 * i + 0] = rel(r64);
  err[4 * i + 1] = rel(newton(r64));
  err[4 * i + 2] = rel(r32);
  err[4 * i + 3] = rel(newton(r32));
}

int main() {
  const int n = 1 << 22;
  std::vector<double> h(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double m = 1.0 + (double)(s >> 11) * (1.0 / 9007199254740992.0);
    const int e = (int)((s >> 3) % 61) - 30;
    h[i] = ldexp(m, e) * ((s & 1) ? 1.0 : -1.0);
  }
  double *dz, *de;
  hipMalloc(&dz, n * sizeof(double));
  hipMalloc(&de, 4 * n * sizeof(double));
  hipMemcpy(dz, h.data(), n * sizeof(double), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(n / 256), dim3(256), 0, 0, dz, n, de);
  std::vector<double> e(4 * n);
  hipMemcpy(e.data(), de, 4 * n * sizeof(double), hipMemcpyDeviceToHost);
  const char* name[4] = {"v_rcp_f64", "v_rcp_f64 + 1 Newton", "v_rcp_f32(float(b))", "v_rcp_f32 + 1 Newton (f64)"};
  for (int k = 0; k < 4; ++k) {
    double mx = 0;
    for (int i = 0; i < n; ++i) mx = fmax(mx, e[4 * i + k]);
    printf("%-28s max relative error %.3e = 2^%.2f\n", name[k], mx, log2(mx));
  }
  return 0;
}
