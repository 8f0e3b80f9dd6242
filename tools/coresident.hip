// Does a small one-workgroup kernel on a second stream start promptly beside a grid that
// keeps every CU full (2 workgroups of 74 KB LDS per CU)?  Prints the small kernel's
// launch-to-finish time alone and beside the big grid, for several LDS footprints and with
// a high-priority stream.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void big(double *out, int iters) {
  extern __shared__ double sh[];
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  if (a == 12345.678) out[0] = a;
}

__global__ __launch_bounds__(256) void small(double *out, int iters) {
  extern __shared__ double sh[];
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  if (a == 12345.678) out[0] = a;
}

int main() {
  double *out;
  CK(hipMalloc(&out, 64));
  int lo, hi;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  printf("priority range: least %d greatest %d\n", lo, hi);
  hipStream_t s1, s2, s3;
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithPriority(&s3, hipStreamNonBlocking, hi));
  CK(hipFuncSetAttribute((const void *)big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int big_iters = 20000;      // ~ 100 us per workgroup
  const int small_iters = 4000;
  for (int lds_kb : {8, 34, 80, 134}) {
    for (int pri = 0; pri < 2; ++pri) {
      hipStream_t ss = pri ? s3 : s2;
      // alone
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(small, dim3(1), dim3(256), lds_kb * 1024, ss, out, small_iters);
      CK(hipStreamSynchronize(ss));
      double alone = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      // beside the big grid: 512 slots x 8 rounds
      hipLaunchKernelGGL(big, dim3(512 * 8), dim3(256), 74 * 1024, s1, out, big_iters);
      std::this_thread::sleep_for(std::chrono::microseconds(300));
      t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(small, dim3(1), dim3(256), lds_kb * 1024, ss, out, small_iters);
      CK(hipStreamSynchronize(ss));
      double beside = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      t0 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s1));
      double rest = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("small LDS %3d KB  priority %s: alone %.0f us, beside %.0f us (big grid ran %.0f us longer)\n",
             lds_kb, pri ? "high" : "same", alone, beside, rest);
    }
  }
  return 0;
}
