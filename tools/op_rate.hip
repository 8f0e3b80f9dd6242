// Throughput of the FP64 vector instructions the ordinal cell is made of, with every SIMD
// kept full (8 independent dependency chains per lane, 8 waves per SIMD): clocks per wave
// instruction and SIMD.  Full-rate FP64 reads 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int ITERS = 4096, CH = 8;

template <int OP>
__global__ __launch_bounds__(256) void rate(double *out, double seed) {
  double v[CH];
  int e[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c) { v[c] = seed + 0.001 * (threadIdx.x + c); e[c] = c; }
  for (int it = 0; it < ITERS; ++it) {
#pragma unroll
    for (int c = 0; c < CH; ++c) {
      if (OP == 0) v[c] = fma(v[c], 1.0000001, 0.5);
      if (OP == 1) v[c] = __builtin_amdgcn_rcp(v[c]);
      if (OP == 2) v[c] = ldexp(v[c], (it & 1) ? 1 : -1);
      if (OP == 3) v[c] = __builtin_amdgcn_frexp_mant(v[c]) + 1.0;
      if (OP == 4) { e[c] += __builtin_amdgcn_frexp_exp(v[c]); v[c] += 1.0; }
      if (OP == 5) v[c] = rint(v[c] * 1.37);
      if (OP == 6) { e[c] += (int)v[c]; v[c] += 1.0; }
      if (OP == 7) v[c] = (double)e[c] + v[c];
      if (OP == 8) v[c] = fmax(v[c], 0.3) * 0.999;
      if (OP == 9) v[c] = v[c] * 0.9999 ;
      if (OP == 10) v[c] = v[c] + 0.9999;
      if (OP == 11) { e[c] = (e[c] >> 3) + it; }
      if (OP == 12) v[c] = log(v[c]) + 3.0;
    }
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CH; ++c) s += v[c] + e[c];
  if (s == 12345.678) out[0] = s;
}

template <int OP>
void run(const char *name, int extra_per_iter, double *out, double clock_ghz) {
  const int wgs = 256 * 8;   // 8 workgroups of 4 waves per CU: 8 waves per SIMD
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  hipLaunchKernelGGL(rate<OP>, dim3(wgs), dim3(256), 0, 0, out, 1.5);
  CK(hipEventRecord(a));
  hipLaunchKernelGGL(rate<OP>, dim3(wgs), dim3(256), 0, 0, out, 1.5);
  CK(hipEventRecord(b));
  CK(hipEventSynchronize(b));
  float ms;
  CK(hipEventElapsedTime(&ms, a, b));
  const double waves_per_simd = wgs * 4.0 / (256 * 4);
  const double instr = (double)ITERS * CH * waves_per_simd;
  printf("%-28s %6.2f clocks per wave instruction (incl. %d helper ops)\n", name,
         ms * 1e-3 * clock_ghz * 1e9 / instr, extra_per_iter);
}

int main() {
  double *out;
  CK(hipMalloc(&out, 64));
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const double ghz = p.clockRate * 1e-6;
  printf("clock %.2f GHz\n", ghz);
  run<0>("v_fma_f64", 0, out, ghz);
  run<9>("v_mul_f64", 0, out, ghz);
  run<10>("v_add_f64", 0, out, ghz);
  run<8>("v_max_f64 + v_mul_f64", 1, out, ghz);
  run<1>("v_rcp_f64", 0, out, ghz);
  run<2>("v_ldexp_f64", 0, out, ghz);
  run<3>("v_frexp_mant_f64 + add", 1, out, ghz);
  run<4>("v_frexp_exp_i32_f64 + 2 adds", 2, out, ghz);
  run<5>("v_rndne_f64 + mul", 1, out, ghz);
  run<6>("v_cvt_i32_f64 + 2 adds", 2, out, ghz);
  run<7>("v_cvt_f64_i32 + add", 1, out, ghz);
  run<11>("int shift + add", 1, out, ghz);
  run<12>("log() + add", 1, out, ghz);
  return 0;
}
