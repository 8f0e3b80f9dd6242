// Relative error of v_rsq_f64 / v_rcp_f64 (hardware estimates) over a spread of arguments.
// Build: hipcc -O3 --offload-arch=gfx950 tools/rsq_err.hip -o tools/rsq_err
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__global__ void k(double *out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double d = exp2(-20.0 + 40.0 * (double)i / n) * (1.0 + 0.37 * ((i * 2654435761u) >> 8) / 16777216.0);
  const double y = __builtin_amdgcn_rsq(d), r = __builtin_amdgcn_rcp(d);
  out[2 * i] = fabs(fma(-d * y, y, 1.0)) * 0.5;      // relative error of y
  out[2 * i + 1] = fabs(fma(-d, r, 1.0));
}
int main() {
  const int n = 1 << 20;
  double *d;
  if (hipMalloc(&d, sizeof(double) * 2 * n) != hipSuccess) return 1;
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, n);
  double *h = new double[2 * n];
  if (hipMemcpy(h, d, sizeof(double) * 2 * n, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  double e0 = 0, e1 = 0;
  for (int i = 0; i < n; ++i) { e0 = fmax(e0, h[2 * i]); e1 = fmax(e1, h[2 * i + 1]); }
  printf("max relative error: v_rsq_f64 %.3e (2^%.1f)   v_rcp_f64 %.3e (2^%.1f)\n", e0, log2(e0), e1, log2(e1));
  return 0;
}
