// How close is rt355::inv_sqrt<double> (hardware reciprocal square root + two Newton steps, kernels.h) to 1 / sqrt(x)?
// One million arguments spread evenly over the exponents 2^-80 .. 2^80, against the long-double value on the host;
// prints the largest error in units of the last place and fails beyond 1.5.  (The expression it replaces -- a square
// root and then a division, each rounded -- is itself good to 1 ulp.)
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=on inv_sqrt_check.hip -o inv_sqrt_check
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "../../raytracer_amd/csrc/kernels.h"

__global__ void k(const double* x, double* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = rt355::inv_sqrt<double>(x[i]);
}

int main() {
  const int n = 1 << 20;
  std::vector<double> x(n), y(n);
  unsigned long long s = 0x355;
  for (int i = 0; i < n; ++i) {
    s = s * 6364136223846793005ull + 1442695040888963407ull;
    const double m = 1.0 + (double)(s >> 11) / 9007199254740992.0;  // [1, 2)
    x[i] = std::ldexp(m, (int)((s >> 3) % 161) - 80);
  }
  double *dx, *dy;
  if (hipMalloc(&dx, n * sizeof(double)) != hipSuccess || hipMalloc(&dy, n * sizeof(double)) != hipSuccess) return 2;
  if (hipMemcpy(dx, x.data(), n * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return 2;
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dy, n);
  if (hipMemcpy(y.data(), dy, n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double worst = 0;
  for (int i = 0; i < n; ++i) {
    const long double want = 1.0L / sqrtl((long double)x[i]);
    const double ulp = std::ldexp(1.0, std::ilogb((double)want) - 52);
    const double err = (double)fabsl((long double)y[i] - want) / ulp;
    if (!(err <= worst)) worst = err;
  }
  std::printf("inv_sqrt<double>: largest error %.3f ulp over %d arguments\n", worst, n);
  (void)hipFree(dx), (void)hipFree(dy);
  return worst <= 1.5 ? 0 : 1;
}
