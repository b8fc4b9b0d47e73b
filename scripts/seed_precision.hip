// accuracy of the v_rcp_f64 / v_rsq_f64 seeds and of one / two refinement steps (decides how many k_spectrum needs)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const double* x, int n, double* err) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double v = x[i];
  const double exact = 1.0 / v;
  double y = __builtin_amdgcn_rcp(v);
  err[0 * n + i] = fabs(y - exact) / exact;
  double e = __builtin_fma(-v, y, 1.0);
  y = __builtin_fma(y, e, y);
  err[1 * n + i] = fabs(y - exact) / exact;
  e = __builtin_fma(-v, y, 1.0);
  y = __builtin_fma(y, e, y);
  err[2 * n + i] = fabs(y - exact) / exact;
  // rsq: sqrt and 1/sqrt after 0, 1, 2 coupled steps (+ the final correction of fsqrt2)
  const double sq = sqrt(v), isq = 1.0 / sq;
  const double r0 = __builtin_amdgcn_rsq(v);
  err[3 * n + i] = fabs(r0 - isq) / isq;
  double g = v * r0, h = 0.5 * r0;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  {
    const double d = __builtin_fma(-g, g, v);
    const double s1 = __builtin_fma(d, h, g);
    err[4 * n + i] = fabs(s1 - sq) / sq;
    err[5 * n + i] = fabs(2 * h - isq) / isq;
  }
  r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  const double d = __builtin_fma(-g, g, v);
  const double s2 = __builtin_fma(d, h, g);
  err[6 * n + i] = fabs(s2 - sq) / sq;
  err[7 * n + i] = fabs(2 * h - isq) / isq;
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n);
  unsigned long long s = 88172645463325252ull;
  for (int i = 0; i < n; ++i) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const double u = (double)(s >> 11) / 9007199254740992.0;
    x[i] = pow(10.0, -30.0 + 60.0 * u);
  }
  double *dx, *de;
  hipMalloc(&dx, n * sizeof(double));
  hipMalloc(&de, 8 * (size_t)n * sizeof(double));
  hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice);
  k<<<n / 256, 256>>>(dx, n, de);
  std::vector<double> e(8 * (size_t)n);
  hipMemcpy(e.data(), de, e.size() * sizeof(double), hipMemcpyDeviceToHost);
  const char* names[8] = {"rcp seed", "rcp + 1 Newton", "rcp + 2 Newton", "rsq seed", "sqrt, 1 step + corr", "1/sqrt, 1 step",
                          "sqrt, 2 steps + corr", "1/sqrt, 2 steps"};
  for (int j = 0; j < 8; ++j) {
    double m = 0;
    for (int i = 0; i < n; ++i) m = fmax(m, e[(size_t)j * n + i]);
    printf("%-22s max rel err %.3e (2^%.1f)\n", names[j], m, log2(m > 0 ? m : 1e-300));
  }
  return 0;
}
