// Single-wave issue-rate probe: clocks per instruction for chains of independent FP64 FMAs
// with 64 or 32 active lanes, FP32 FMAs, and LDS broadcast reads.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(long long *out, double *sink, int lanes) {
  __shared__ double sh[64];
  sh[threadIdx.x] = threadIdx.x;
  __syncthreads();
  double a[16];
  float f[16];
  for (int i = 0; i < 16; ++i) { a[i] = threadIdx.x + i; f[i] = threadIdx.x + i; }
  double m = 1.0000001;
  long long t[6];
  if ((int)threadIdx.x < lanes) {
    t[0] = clock64();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { a[i] = a[i] * m + 0.5; asm volatile("" : "+v"(a[i])); }
    }
    t[1] = clock64();
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { f[i] = f[i] * 1.0001f + 0.5f; asm volatile("" : "+v"(f[i])); }
    }
    t[2] = clock64();
    double acc = 0;
    for (int it = 0; it < 64; ++it) {
#pragma unroll
      for (int i = 0; i < 16; ++i) { double v = sh[(it + i) & 63]; asm volatile("" : "+v"(v)); acc += v; }
    }
    t[3] = clock64();
    // dependent chain
    double c = a[0];
    for (int it = 0; it < 1024; ++it) { c = c * m + 0.5; asm volatile("" : "+v"(c)); }
    t[4] = clock64();
    if (threadIdx.x == 0) { out[0] = t[1] - t[0]; out[1] = t[2] - t[1]; out[2] = t[3] - t[2]; out[3] = t[4] - t[3]; }
    double s = acc + c;
    for (int i = 0; i < 16; ++i) s += a[i] + f[i];
    sink[threadIdx.x] = s;
  }
}

int main() {
  long long *out; double *sink;
  if (hipMalloc(&out, 64) != hipSuccess || hipMalloc(&sink, 64 * 8) != hipSuccess) return 1;
  for (int lanes : {64, 32, 16}) {
    long long h[4];
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, sink, lanes);
    if (hipMemcpy(h, out, 32, hipMemcpyDeviceToHost) != hipSuccess) return 2;
    printf("lanes=%d  f64 fma %.2f clk/instr  f32 fma %.2f  lds read+add %.2f per pair  dependent f64 fma %.2f\n",
           lanes, h[0] / 1024.0, h[1] / 1024.0, h[2] / 1024.0, h[3] / 1024.0);
  }
  return 0;
}
