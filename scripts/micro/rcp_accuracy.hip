// development aid: how accurate is a bare v_rcp_f64 on gfx950 (no Newton step)? max / mean relative error against 1/x in
// long double over mantissas in [1, 2) and over a wide range of exponents.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
__global__ void k(const double* x, double* r0, double* r1, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double q = __builtin_amdgcn_rcp(x[i]);
  r0[i] = q;
  r1[i] = fma(q, fma(-x[i], q, 1.0), q);
}
int main() {
  const int n = 1 << 22;
  std::vector<double> x(n), a(n), b(n);
  std::mt19937_64 g(1);
  std::uniform_real_distribution<double> u(1.0, 2.0), e(-300.0, 300.0);
  for (int i = 0; i < n; ++i) x[i] = i < n / 2 ? u(g) : u(g) * std::exp2(std::floor(e(g)));
  double *dx, *d0, *d1;
  hipMalloc(&dx, n * 8); hipMalloc(&d0, n * 8); hipMalloc(&d1, n * 8);
  hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, d0, d1, n);
  hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost); hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost);
  for (int part = 0; part < 2; ++part) {
    long double m0 = 0, m1 = 0, s0 = 0, s1 = 0;
    for (int i = part * n / 2; i < (part + 1) * n / 2; ++i) {
      const long double t = 1.0L / (long double)x[i];
      const long double e0 = fabsl(((long double)a[i] - t) / t), e1 = fabsl(((long double)b[i] - t) / t);
      if (e0 > m0) m0 = e0; if (e1 > m1) m1 = e1; s0 += e0; s1 += e1;
    }
    printf("%s: bare v_rcp_f64 max rel err %.3Le (%.2Lf ulp of 2^-53), mean %.3Le; with one Newton step max %.3Le, mean %.3Le\n",
           part ? "wide exponents" : "[1,2)", m0, m0 / 1.1102230246251565e-16L, s0 / (n / 2), m1, s1 / (n / 2));
  }
  return 0;
}
