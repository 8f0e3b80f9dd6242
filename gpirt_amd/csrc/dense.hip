// Dense FP64 pieces of the sweep for gfx950: kernel-matrix build, blocked Cholesky,
// triangular products/solves -- all on one MFMA GEMM core (v_mfma_f64_16x16x4_f64).
//
//   K(theta,theta)+jitter  main.m:29,67 (covSEiso, utils/model.m:4,6); getprob_gpirt.R:41-51
//   chol                   utils/sample_fs.m:2 (inside mvnrnd); generate_data.R:56
//   nu = L z               utils/elliptical_slice.m:64
//   S^-1 B                 the solve inside gp(...), main.m:57-58
#include "common.h"

namespace gpirt {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BK = 16;        // k-slab depth held in LDS
constexpr int LDS_LD = 144;   // 128 + 16: the four k-rows one ds_read_b64 touches land on
                              // disjoint bank halves

// ---------------------------------------------------------------------------------------
// GEMM core.  One 256-thread workgroup owns a 128x128 tile of C; its four waves own 64x64
// quadrants as 4x4 MFMA tiles of 16x16.  The MFMA is fed transposed (its "A" operand is
// the B-tile, its "B" operand the A-tile) so that the 16 lanes sharing an accumulator
// register hold 16 consecutive rows i of one column j: the epilogue then writes 128-byte
// runs of the column-major C instead of 8-byte scatters.
// ---------------------------------------------------------------------------------------
template <bool KCONTIG>
__device__ inline void tile_load(const double *__restrict__ base, int64_t ld, int row0, int k0,
                                 int tid, double2 (&r)[4]) {
  if (!KCONTIG) {  // element (row,k) at base[row + ld*k]: rows contiguous
    int kl = tid >> 6, r2 = (tid & 63) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      r[p] = *reinterpret_cast<const double2 *>(base + (row0 + r2) + ld * (int64_t)(k0 + kl + 4 * p));
  } else {         // element (row,k) at base[k + ld*row]: k contiguous
    int rl = tid >> 3, k2 = (tid & 7) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p)
      r[p] = *reinterpret_cast<const double2 *>(base + (k0 + k2) + ld * (int64_t)(row0 + rl + 32 * p));
  }
}

template <bool KCONTIG>
__device__ inline void tile_store(double (*s)[LDS_LD], int tid, const double2 (&r)[4]) {
  if (!KCONTIG) {
    int kl = tid >> 6, r2 = (tid & 63) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<double2 *>(&s[kl + 4 * p][r2]) = r[p];
  } else {
    int rl = tid >> 3, k2 = (tid & 7) * 2;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      s[k2][rl + 32 * p] = r[p].x;
      s[k2 + 1][rl + 32 * p] = r[p].y;
    }
  }
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256) void gemm_f64_kernel(GemmArgs g, int nbm, int nbn) {
  __shared__ double sA[2][BK][LDS_LD];
  __shared__ double sB[2][BK][LDS_LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x, bi, bj;
  if (g.lower_only) {
    bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while ((int64_t)bi * (bi + 1) / 2 > t) --bi;
    bj = t - (int)((int64_t)bi * (bi + 1) / 2);
  } else if (g.k_trim == 1) {   // longest K first
    bi = nbm - 1 - t / nbn; bj = t % nbn;
  } else {
    bi = t / nbn; bj = t % nbn;
  }
  const int z = blockIdx.y;
  const double *A = g.A + g.strideA * z;
  const double *B = g.B + g.strideB * z;
  double *C = g.C + g.strideC * z;
  const int i0 = bi * TILE, j0 = bj * TILE;
  int kbeg = 0, kend = g.K;
  if (g.k_trim == 1) kend = min(g.K, (bi + 1) * TILE);
  if (g.k_trim == 2) kbeg = bi * TILE;

  v4d acc[4][4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};

  double2 ra[4], rb[4];
  tile_load<AT>(A, g.lda, i0, kbeg, tid, ra);
  tile_load<!BT>(B, g.ldb, j0, kbeg, tid, rb);
  tile_store<AT>(sA[0], tid, ra);
  tile_store<!BT>(sB[0], tid, rb);
  __syncthreads();

  const int wm0 = (wave & 1) * 64, wn0 = (wave >> 1) * 64;
  const int lr = lane & 15, lk = lane >> 4;
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const bool more = k0 + BK < kend;
    if (more) {
      tile_load<AT>(A, g.lda, i0, k0 + BK, tid, ra);
      tile_load<!BT>(B, g.ldb, j0, k0 + BK, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      double fa[4], fb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        fa[q] = sB[buf][kk * 4 + lk][wn0 + q * 16 + lr];  // MFMA row index  = j
        fb[q] = sA[buf][kk * 4 + lk][wm0 + q * 16 + lr];  // MFMA col index  = i
      }
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[tj], fb[ti], acc[tj][ti], 0, 0, 0);
    }
    if (more) {
      tile_store<AT>(sA[buf ^ 1], tid, ra);
      tile_store<!BT>(sB[buf ^ 1], tid, rb);
    }
    __syncthreads();
    buf ^= 1;
  }

  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = j0 + wn0 + tj * 16 + lk + 4 * r;
        int i = i0 + wm0 + ti * 16 + lr;
        double *c = C + i + g.ldc * (int64_t)j;
        double v = alpha * acc[tj][ti][r];
        if (beta != 0.0) v += beta * *c;
        *c = v;
      }
}

int launch_gemm(hipStream_t st, bool AT, bool BT, const GemmArgs &a) {
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return GPIRT_OK;
  if (a.M % TILE || a.N % TILE || a.K % BK || a.K <= 0) {
    set_error("launch_gemm: M=%d N=%d K=%d not tile multiples", a.M, a.N, a.K);
    return GPIRT_ERR_BAD_ARG;
  }
  int nbm = a.M / TILE, nbn = a.N / TILE;
  int64_t tiles = a.lower_only ? (int64_t)nbm * (nbm + 1) / 2 : (int64_t)nbm * nbn;
  dim3 grid((unsigned)tiles, (unsigned)a.batch), block(256);
  if (!AT && !BT) hipLaunchKernelGGL((gemm_f64_kernel<false, false>), grid, block, 0, st, a, nbm, nbn);
  else if (!AT && BT) hipLaunchKernelGGL((gemm_f64_kernel<false, true>), grid, block, 0, st, a, nbm, nbn);
  else if (AT && !BT) hipLaunchKernelGGL((gemm_f64_kernel<true, false>), grid, block, 0, st, a, nbm, nbn);
  else hipLaunchKernelGGL((gemm_f64_kernel<true, true>), grid, block, 0, st, a, nbm, nbn);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// Kernel-matrix build (a1) and the grid cross-covariance K(x, xs) used by f* (a6)
// ---------------------------------------------------------------------------------------
__global__ void k_build_kernel(const double *__restrict__ x, int64_t x_stride, int n, int n_pad,
                               double inv_ls2, double sf2, double jitter, double *__restrict__ S,
                               int full) {
  const int z = blockIdx.z;
  const double *xp = x + x_stride * z;
  double *Sp = S + (int64_t)n_pad * n_pad * z;
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (i >= n_pad) return;
  double v;
  if (i < n && j < n) {
    if (i >= j || full) {
      double d = xp[i] - xp[j];
      v = sf2 * exp(-0.5 * d * d * inv_ls2) + (i == j ? jitter : 0.0);
    } else v = 0.0;
  } else v = (i == j) ? 1.0 : 0.0;
  Sp[i + (int64_t)n_pad * j] = v;
}

int launch_k_build(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                   double ls, double sf, double jitter, double *S, bool full) {
  dim3 block(256), grid((n_pad + 255) / 256, n_pad, batch);
  hipLaunchKernelGGL(k_build_kernel, grid, block, 0, st, x, x_stride, n, n_pad, 1.0 / (ls * ls),
                     sf * sf, jitter, S, full ? 1 : 0);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

__global__ void kstar_t_kernel(const double *__restrict__ x, int64_t x_stride, int n, int n_pad,
                               double inv_ls2, double sf2, int G, double lo, double step,
                               double *__restrict__ Kst, int G_pad) {
  const int z = blockIdx.z;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  int gidx = blockIdx.y;
  if (r >= n_pad) return;
  double v = 0.0;
  if (r < n && gidx < G) {
    double d = (lo + step * gidx) - x[x_stride * z + r];
    v = sf2 * exp(-0.5 * d * d * inv_ls2);
  }
  Kst[(int64_t)n_pad * G_pad * z + r + (int64_t)n_pad * gidx] = v;
}

int launch_kstar_t(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                   double ls, double sf, int G, int G_pad, double lo, double step, double *Kst) {
  dim3 block(256), grid((n_pad + 255) / 256, G_pad, batch);
  hipLaunchKernelGGL(kstar_t_kernel, grid, block, 0, st, x, x_stride, n, n_pad, 1.0 / (ls * ls),
                     sf * sf, G, lo, step, Kst, G_pad);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// Diagonal-block factorisation: one workgroup factors a 128x128 block held in LDS and
// inverts the factor.  The inverse (lower triangular) is built transposed in the strict
// upper triangle of the same LDS image, so one 132 KB array holds both.
// ---------------------------------------------------------------------------------------
constexpr int DLD = TILE + 1;

__global__ __launch_bounds__(256) void potrf_diag_kernel(double *__restrict__ S, int n_pad, int kb,
                                                        double *__restrict__ inv_diag, int nb,
                                                        int *__restrict__ info) {
  __shared__ double T[TILE][DLD];
  __shared__ double col[TILE];
  __shared__ double xd[TILE];
  const int z = blockIdx.x, tid = threadIdx.x;
  double *A = S + (int64_t)n_pad * n_pad * z + (int64_t)kb * TILE * (n_pad + 1);
  double *Inv = inv_diag + (int64_t)TILE * TILE * (kb + (int64_t)nb * z);

  for (int e = tid; e < TILE * TILE; e += 256) {
    int i = e & (TILE - 1), j = e >> 7;
    T[i][j] = (i >= j) ? A[i + (int64_t)n_pad * j] : 0.0;
  }
  __syncthreads();

  const int row = tid & (TILE - 1), half = tid >> 7;
  for (int j = 0; j < TILE; ++j) {
    double d = T[j][j];
    if (!(d > 0.0)) {
      if (tid == 0) atomicMax(info, kb * TILE + j + 1);
      d = 1.0;
    }
    double rs = 1.0 / sqrt(d);
    if (half == 0 && row >= j) col[row] = T[row][j] * rs;
    __syncthreads();
    if (half == 0 && row >= j) T[row][j] = col[row];
    const double ci = col[row];
    for (int c = j + 1 + half; c <= row; c += 2) T[row][c] -= ci * col[c];
    __syncthreads();
  }

  // inverse of the factor, column t by thread t, stored transposed: X[i][t] -> T[t][i] (i>t)
  if (tid < TILE) {
    const int t = tid;
    xd[t] = 1.0 / T[t][t];
    for (int i = t + 1; i < TILE; ++i) {
      double s = T[i][t] * xd[t];
      for (int c = t + 1; c < i; ++c) s += T[i][c] * T[t][c];
      T[t][i] = -s / T[i][i];
    }
  }
  __syncthreads();

  for (int e = tid; e < TILE * TILE; e += 256) {
    int i = e & (TILE - 1), j = e >> 7;
    A[i + (int64_t)n_pad * j] = (i >= j) ? T[i][j] : 0.0;
    Inv[i + TILE * j] = (i > j) ? T[j][i] : (i == j ? xd[i] : 0.0);
  }
}

int launch_potrf(hipStream_t st, double *S, int n_pad, int batch, double *inv_diag, int *d_info) {
  const int nb = n_pad / TILE;
  const int64_t sS = (int64_t)n_pad * n_pad, sI = (int64_t)TILE * TILE * nb;
  for (int k = 0; k < nb; ++k) {
    hipLaunchKernelGGL(potrf_diag_kernel, dim3(batch), dim3(256), 0, st, S, n_pad, k, inv_diag, nb,
                       d_info);
    GP_HIP(hipGetLastError());
    int rem = nb - 1 - k;
    if (!rem) break;
    double *panel = S + (int64_t)(k + 1) * TILE + (int64_t)n_pad * k * TILE;
    GemmArgs p{};  // L_ik = A_ik * inv(L_kk)^T
    p.A = panel; p.lda = n_pad; p.strideA = sS;
    p.B = inv_diag + (int64_t)TILE * TILE * k; p.ldb = TILE; p.strideB = sI;
    p.C = panel; p.ldc = n_pad; p.strideC = sS;
    p.M = rem * TILE; p.N = TILE; p.K = TILE; p.batch = batch; p.alpha = 1.0; p.beta = 0.0;
    GP_TRY(launch_gemm(st, false, true, p));
    GemmArgs u{};  // A_ij -= L_ik L_jk^T on the lower tiles
    u.A = panel; u.lda = n_pad; u.strideA = sS;
    u.B = panel; u.ldb = n_pad; u.strideB = sS;
    u.C = S + (int64_t)(k + 1) * TILE * (n_pad + 1); u.ldc = n_pad; u.strideC = sS;
    u.M = rem * TILE; u.N = rem * TILE; u.K = TILE; u.batch = batch; u.alpha = -1.0; u.beta = 1.0;
    u.lower_only = 1;
    GP_TRY(launch_gemm(st, false, true, u));
  }
  return GPIRT_OK;
}

int launch_potrs(hipStream_t st, const double *L, const double *inv_diag, int n_pad, int batch,
                 double *B, int64_t ldb, int64_t strideB, int nrhs_pad) {
  const int nb = n_pad / TILE;
  const int64_t sS = (int64_t)n_pad * n_pad, sI = (int64_t)TILE * TILE * nb;
  for (int k = 0; k < nb; ++k) {  // L W = B
    GemmArgs d{};
    d.A = inv_diag + (int64_t)TILE * TILE * k; d.lda = TILE; d.strideA = sI;
    d.B = B + (int64_t)k * TILE; d.ldb = ldb; d.strideB = strideB;
    d.C = B + (int64_t)k * TILE; d.ldc = ldb; d.strideC = strideB;
    d.M = TILE; d.N = nrhs_pad; d.K = TILE; d.batch = batch; d.alpha = 1.0; d.beta = 0.0;
    GP_TRY(launch_gemm(st, false, false, d));
    int rem = nb - 1 - k;
    if (!rem) break;
    GemmArgs u{};
    u.A = L + (int64_t)(k + 1) * TILE + (int64_t)n_pad * k * TILE; u.lda = n_pad; u.strideA = sS;
    u.B = B + (int64_t)k * TILE; u.ldb = ldb; u.strideB = strideB;
    u.C = B + (int64_t)(k + 1) * TILE; u.ldc = ldb; u.strideC = strideB;
    u.M = rem * TILE; u.N = nrhs_pad; u.K = TILE; u.batch = batch; u.alpha = -1.0; u.beta = 1.0;
    GP_TRY(launch_gemm(st, false, false, u));
  }
  for (int k = nb - 1; k >= 0; --k) {  // L' X = W
    GemmArgs d{};
    d.A = inv_diag + (int64_t)TILE * TILE * k; d.lda = TILE; d.strideA = sI;
    d.B = B + (int64_t)k * TILE; d.ldb = ldb; d.strideB = strideB;
    d.C = B + (int64_t)k * TILE; d.ldc = ldb; d.strideC = strideB;
    d.M = TILE; d.N = nrhs_pad; d.K = TILE; d.batch = batch; d.alpha = 1.0; d.beta = 0.0;
    GP_TRY(launch_gemm(st, true, false, d));
    if (!k) break;
    GemmArgs u{};  // B_i -= L_ki^T B_k for i < k
    u.A = L + (int64_t)k * TILE; u.lda = n_pad; u.strideA = sS;
    u.B = B + (int64_t)k * TILE; u.ldb = ldb; u.strideB = strideB;
    u.C = B; u.ldc = ldb; u.strideC = strideB;
    u.M = k * TILE; u.N = nrhs_pad; u.K = TILE; u.batch = batch; u.alpha = -1.0; u.beta = 1.0;
    GP_TRY(launch_gemm(st, true, false, u));
  }
  return GPIRT_OK;
}

}  // namespace gpirt
