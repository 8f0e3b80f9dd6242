// Measures the sustained rate of v_mfma_f64_16x16x4_f64 from registers only (no memory
// traffic): the practical ceiling of the FP64 matrix pipe at the clock the chip holds under
// this load.  Build: hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void mfma_loop(double *out, int iters, double a0, double b0) {
  v4d acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main(int argc, char **argv) {
  const int wgs = argc > 1 ? atoi(argv[1]) : 512, iters = argc > 2 ? atoi(argv[2]) : 20000;
  double *out;
  if (hipMalloc(&out, sizeof(double) * wgs * 256) != hipSuccess) return 1;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(mfma_loop, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0, 1e-3);
    hipEventRecord(e1, 0);
    if (hipEventSynchronize(e1) != hipSuccess) return 2;
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    double flop = (double)wgs * 4 * iters * 16 * 2048.0;
    printf("wgs=%d iters=%d  %.3f ms  %.2f TFLOP/s\n", wgs, iters, ms, flop / ms * 1e-9);
  }
  hipFree(out);
  return 0;
}
