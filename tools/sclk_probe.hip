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

// the same loop with the address arithmetic taken out of it: pointers that advance by a block, loads through
// raw buffer resources (scalar base + 32-bit lane offset), the last blocks drained without prefetch
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ double2 buf_ld(rsrc_t rsrc, uint32_t voff, uint32_t soff) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u r = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
  double2 d; __builtin_memcpy(&d, &r, 16); return d;
}
__global__ __launch_bounds__(256, 2) void gemm8p(const double *__restrict__ A, const double *__restrict__ B, double *__restrict__ C,
                                                  int K, int64_t lda, int64_t ldb, int64_t ldc, int nbn) {
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lr = lane & 15, lk = lane >> 4;
  const int bi = blockIdx.x / nbn, bj = blockIdx.x % nbn;
  const int i0 = bi * 128 + (wave & 1) * 64, j0 = bj * 128 + (wave >> 1) * 64;
  // resources at the wave's tile; a block of 8 k-values is lda * 64 bytes further in A and 64 bytes further in B
  const rsrc_t ra = make_rsrc(A + i0), rb = make_rsrc(B + ldb * (int64_t)j0);
  uint32_t oa[2], ob[4];
  for (int s = 0; s < 2; ++s) oa[s] = (uint32_t)((2 * lr + lda * (2 * lk + s)) * 8);
  for (int tj = 0; tj < 4; ++tj) ob[tj] = (uint32_t)((2 * lk + ldb * (16 * tj + lr)) * 8);
  const uint32_t stepa = (uint32_t)(lda * 64), stepb = 64;
  v4d acc[4][4];
  for (int a = 0; a < 4; ++a) for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0, 0, 0, 0};
  constexpr int NB = 3;
  double fa[NB][2][4], fb[NB][2][4];
  const int nblk = K / 8;
  uint32_t sa = 0, sb = 0;      // scalar byte offsets of the next block to load
  int next = 0;                 // its number; past the last block the last one is loaded again
  auto load = [&](int buf) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const double2 v = buf_ld(ra, oa[s] + 256 * g, sa);
        fa[buf][s][2 * g] = v.x; fa[buf][s][2 * g + 1] = v.y;
      }
#pragma unroll
    for (int tj = 0; tj < 4; ++tj) {
      const double2 v = buf_ld(rb, ob[tj], sb);
      fb[buf][0][tj] = v.x; fb[buf][1][tj] = v.y;
    }
    ++next;
    const bool more = next < nblk;
    sa += more ? stepa : 0; sb += more ? stepb : 0;
  };
  auto mm = [&](int u) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[u][s][tj], fa[u][s][ti], acc[tj][ti], 0, 0, 0);
  };
  load(0); load(1);
  const int nmain = nblk / NB * NB;
  for (int blk = 0; blk < nmain; blk += NB)
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      load((u + NB - 1) % NB);
      __builtin_amdgcn_sched_barrier(0);
      mm(u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
  for (int u = 0; u < NB - 1; ++u)
    if (nmain + u < nblk) mm(u);
  for (int tj = 0; tj < 4; ++tj) for (int g = 0; g < 2; ++g) for (int r = 0; r < 4; ++r)
    *reinterpret_cast<double2 *>(C + i0 + 32 * g + 2 * lr + ldc * (int64_t)(j0 + 16 * tj + lk + 4 * r)) =
        make_double2(acc[tj][2 * g][r], acc[tj][2 * g + 1][r]);
}

int main(int argc, char **argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;        // 1 / 2: that load alone, for argv[2] seconds (sample rocm-smi beside it)
  const double seconds = argc > 2 ? atof(argv[2]) : 3.0;
  const int kmask = argc > 3 ? atoi(argv[3]) : -1;
  const int lds = argc > 4 ? atoi(argv[4]) : 0;          // dynamic LDS per workgroup: > 80 KB keeps a CU to one workgroup
  if (lds > 65536) {
    CK(hipFuncSetAttribute((const void *)mfma_only, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CK(hipFuncSetAttribute((const void *)gemm8p, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
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
      if (only == 1) hipLaunchKernelGGL(mfma_only, dim3(2048), dim3(256), lds, s1, 13000, C);
      else if (only == 3) hipLaunchKernelGGL(gemm8p, dim3((M / 128) * (N / 128)), dim3(256), lds, s1, A, B, C, K, (int64_t)M, (int64_t)K, (int64_t)M, N / 128);
      else hipLaunchKernelGGL(gemm8, dim3((M / 128) * (N / 128)), dim3(256), 0, s1, A, B, C, K, (int64_t)M, (int64_t)K, (int64_t)M, N / 128, kmask);
    }
    CK(hipEventRecord(e1, s1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = (only == 1 ? 4.0 * 2048 * 13000.0 * 16 * 2048 : 2.0 * M * N * K) * reps;
    printf("%s for %.2f s: %.1f TFLOP/s\n", only == 3 ? "GEMM, buffer loads + pointer steps" : names[only], ms * 1e-3, flop / (ms * 1e-3) * 1e-12);
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
