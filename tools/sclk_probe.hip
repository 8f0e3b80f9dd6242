// What clock does the chip hold under an FP64 MFMA GEMM?  A one-wave kernel on a second stream reads the
// shader cycle counter and the constant 100 MHz counter around a fixed stretch of wall time (1 ms) while
// the GEMM of tools/gemm_direct.hip (same loop: 64 x 64 per wave from registers fed by global loads) or
// a pure register-fed MFMA loop runs on the first; idle = nothing beside it.
// Build: hipcc -O3 --offload-arch=gfx950 tools/sclk_probe.hip -o tools/sclk_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void probe_kernel(long long wall_ticks, long long *out) {
  const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  while (wall_clock64() - w0 < wall_ticks) __builtin_amdgcn_s_sleep(8);
  const long long w1 = wall_clock64(), c1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; }
}

// MFMA from registers only: 16 accumulators a wave, `iters` rounds of 16 MFMAs
__global__ __launch_bounds__(256, 2) void mfma_only(int iters, double *out) {
  v4d acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = (v4d){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 - a;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1234.5) out[0] = s;
}

// the register-fed GEMM loop (A row-contiguous, B K-contiguous), 128 x 128 tile a workgroup
__global__ __launch_bounds__(256, 2) void gemm8(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C,
                                                 int K, int64_t lda, int64_t ldb, int64_t ldc, int nbn, int kmask) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, lk = lane >> 4;
  const int bi = blockIdx.x / nbn, bj = blockIdx.x % nbn;
  const int i0 = bi * 128 + (wave & 1) * 64, j0 = bj * 128 + (wave >> 1) * 64;
  const double *Ap = A + (i0 + 2 * lr) + lda * (int64_t)(2 * lk);
  const double *Bp = B + (2 * lk) + ldb * (int64_t)(j0 + lr);
  v4d acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0, 0, 0, 0};
  constexpr int NB = 3;
  double fa[NB][2][4], fb[NB][2][4];
  const int nblk = K / 8;
  auto load = [&](int buf, int blk) {
    const int k0 = 8 * ((blk < nblk ? blk : nblk - 1) & kmask);     // kmask = 3: the same four blocks over and over (L1 hits)
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const double2 v = *reinterpret_cast<const double2 *>(Ap + 32 * g + lda * (int64_t)(k0 + s));
        fa[buf][s][2 * g] = v.x; fa[buf][s][2 * g + 1] = v.y;
      }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      const double2 v = *reinterpret_cast<const double2 *>(Bp + k0 + ldb * (int64_t)(16 * tj));
      fb[buf][0][tj] = v.x; fb[buf][1][tj] = v.y;
    }
  };
  load(0, 0); load(1, 1);
  for (int blk = 0; blk < nblk; blk += NB)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      load((u + NB - 1) % NB, blk + u + NB - 1);
      __builtin_amdgcn_sched_barrier(0);
      if (blk + u < nblk)
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int tj = 0; tj < 4; ++tj)
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
              acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[u][s][tj], fa[u][s][ti], acc[tj][ti], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  for (int tj = 0; tj < 4; ++tj) for (int g = 0; g < 2; ++g) for (int r = 0; r < 4; ++r)
    *reinterpret_cast<double2 *>(C + i0 + 32 * g + 2 * lr + ldc * (int64_t)(j0 + 16 * tj + lk + 4 * r)) =
        make_double2(acc[tj][2 * g][r], acc[tj][2 * g + 1][r]);
}

int main(int argc, char **argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;        // 1 / 2: that load alone, for argv[2] seconds (sample rocm-smi beside it)
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  const int kmask = argc > 3 ? atoi(argv[3]) : -1;
  const int M = 9600, N = 4096, K = 4800;
  double *A, *B, *C; long long *out;
  CK(hipMalloc(&A, sizeof(double) * (size_t)M * K)); CK(hipMalloc(&B, sizeof(double) * (size_t)K * N));
  CK(hipMalloc(&C, sizeof(double) * (size_t)M * N)); CK(hipMalloc(&out, 64));
  CK(hipMemset(A, 0, sizeof(double) * (size_t)M * K)); CK(hipMemset(B, 0, sizeof(double) * (size_t)K * N));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char *names[3] = {"idle", "MFMA from registers", "register-fed GEMM"};
  if (only > 0) {
    const int reps = (int)(seconds / (only == 1 ? 0.045 : 0.006)) + 1;
    CK(hipEventRecord(e0, s1));
    for (int q = 0; q < reps; ++q) {
      if (only == 1) hipLaunchKernelGGL(mfma_only, dim3(2048), dim3(256), 0, s1, 13000, C);
      else hipLaunchKernelGGL(gemm8, dim3((M / 128) * (N / 128)), dim3(256), 0, s1, A, B, C, K, (int64_t)M, (int64_t)K, (int64_t)M, N / 128, kmask);
    }
    CK(hipEventRecord(e1, s1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (only == 1 ? 4.0 * 2048 * 13000.0 * 16 * 2048 : 2.0 * M * N * K) * reps;
    printf("%s for %.2f s: %.1f TFLOP/s\n", names[only], ms * 1e-3, flop / (ms * 1e-3) * 1e-12);
    return 0;
  }
  for (int mode = 0; mode < 3; ++mode)
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0, s1));
      for (int q = 0; q < 4; ++q) {     // ~20 ms of load
        if (mode == 1) hipLaunchKernelGGL(mfma_only, dim3(2048), dim3(256), 0, s1, 13000, C);
        if (mode == 2) hipLaunchKernelGGL(gemm8, dim3((M / 128) * (N / 128)), dim3(256), 0, s1, A, B, C, K, (int64_t)M, (int64_t)K, (int64_t)M, N / 128, -1);
      }
      CK(hipEventRecord(e1, s1));
      hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, s2, 100000LL * 8, out);     // 8 ms of wall time, started beside the load
      CK(hipDeviceSynchronize());
      long long h[2]; CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      const double flop = mode == 1 ? 4.0 * 2048 * 4 * 13000.0 * 16 * 2048 : mode == 2 ? 4.0 * 2.0 * M * N * K : 0;
      printf("%-22s load %.2f ms  %.1f TFLOP/s   probe: %lld cycles in %lld ticks of 100 MHz = %.3f GHz\n", names[mode], ms,
             flop / (ms * 1e-3) * 1e-12, h[0], h[1], h[0] / (h[1] * 10.0) );
    }
  return 0;
}
