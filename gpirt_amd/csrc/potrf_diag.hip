// Diagonal-block kernel of the blocked Cholesky (chol inside mvnrnd, utils/sample_fs.m:2):
// factor one 128x128 block and invert the factor.
#include "common.h"
#include <cstdlib>

namespace gpirt {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int XLD = TILE + 2;   // row stride of the LDS image (even: pairs stay 16-byte aligned)
constexpr int SB = 32;          // inner block size

#ifdef DIAG_STAMPS
__device__ long long g_diag_stamps[32];
#define STAMP(i) do { if (threadIdx.x == 0) g_diag_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#define STAMP1(i) do { if (threadIdx.x == 64) g_diag_stamps[i] = (long long)__builtin_readcyclecounter(); } while (0)
#else
#define STAMP(i) do { } while (0)
#define STAMP1(i) do { } while (0)
#endif

__device__ __forceinline__ v4d mfma4(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// 1/sqrt(d): hardware estimate (relative error 2^-24, tools/rsq_err.hip) plus ONE third-order
// step: with e = 1 - d y^2, 1/sqrt(d) = y (1 + e/2 + 3 e^2/8 + O(e^3)), remainder 2^-70.  Four
// dependent operations where two Newton steps are eight: this sits in the pivot loop's chain.
__device__ __forceinline__ double rsqrt_nr(double d) {
  const double y = __builtin_amdgcn_rsq(d);
  const double e = fma(-(d * y), y, 1.0);
  return fma(y * e, fma(e, 0.375, 0.5), y);
}

// The inverse X = L^-1 is kept transposed in the strict upper triangle of the same LDS image
// (X[r][c], r > c, at T[c][r]) with its diagonal in dinv: one 133 KB image holds L and X.
__device__ __forceinline__ double x_at(const double *T, const double *dinv, int r, int c) {
  return r > c ? T[c * XLD + r] : (r == c ? dinv[r] : 0.0);
}
// the same element with both loads issued unconditionally (always in bounds) and a select:
// no divergent control flow inside the MFMA loops
__device__ __forceinline__ double x_sel(const double *T, const double *dinv, int r, int c) {
  const double t = T[c * XLD + r], dd = dinv[r];
  return r > c ? t : (r == c ? dd : 0.0);
}

// ---- inner 32x32 block on one wave ----------------------------------------------------------
// Lanes 0..31 keep the rows of the block (lane r: row r of A, becoming row r of L); lanes
// 32..63 keep the columns of B, which starts as the identity and ends as X = L^-1 (forward
// substitution L X = I, column-oriented).  Pivot step J of the factorisation and step J of
// the substitution are the same arithmetic on the two halves: with the pivot column
// uc[c] = A[c][J] (c > J, before scaling) and y = 1/sqrt(A[J][J]),
//     x[c] -= (x[J] y^2) uc[c],   x[J] *= y
// gives L[r][J] = a[J] y on the low half and X[J][t] = b[J] y on the high half, so the
// inverse costs no extra instructions -- a single wave issues one instruction per ~4.5 clocks
// whatever the lane count, and that issue rate is what bounds this loop.  The pivot column
// travels through LDS (ucol); the next step's column is final after this step's first
// update, so it is exchanged right then (into un[]) and its round trip overlaps the rest of
// the step and the next 1/sqrt chain.
template <int J>
__device__ __forceinline__ void blk_step(double (&x)[SB], double (&uc)[SB], double (&un)[SB],
                                         double *ucol, double *dinv_p, int lane, int &bad) {
  // the pivot comes straight out of lane J's register
  double d = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x[J]), J),
                              __builtin_amdgcn_readlane(__double2loint(x[J]), J));
  if (!(d > 0.0)) { bad = J + 1; d = 1.0; }
  asm volatile("" : "+v"(bad));         // settle the flag now: 32 deferred conditions spilled SGPRs
  const double y = rsqrt_nr(d);
  const double w = x[J] * (y * y);        // A[r][J] / d   |   X[J][t] / L[J][J]
  if constexpr (J + 1 < SB) {
    x[J + 1] -= w * uc[J + 1];
    asm volatile("" : "+v"(x[J + 1]));
    ucol[lane] = x[J + 1];                // wave-local exchange (LDS ops of a wave stay in order);
                                          // the high half's 32 entries are never read
#pragma unroll
    for (int c = J + 2; c < SB; ++c) un[c] = ucol[c];
    __builtin_amdgcn_sched_barrier(0);
  }
  x[J] = x[J] * y;                        // L[r][J]       |   X[J][t]
  dinv_p[J] = y;                          // every lane holds the same y: one plain store
#pragma unroll
  for (int c = J + 2; c < SB; ++c) {
    x[c] -= w * uc[c];
    // pin the update to this step: left alone, the compiler defers every update that the
    // next pivot does not need and keeps w and the pivot column of all 32 steps alive
    asm volatile("" : "+v"(x[c]));
  }
}
template <int J0, int J1>
__device__ __forceinline__ void blk_range(double (&x)[SB], double (&uc)[SB], double (&un)[SB],
                                          double *ucol, double *dinv_p, int lane, int &bad) {
  if constexpr (J0 < J1) {
    blk_step<J0>(x, uc, un, ucol, dinv_p, lane, bad);
    blk_range<J0 + 1, J1>(x, un, uc, ucol, dinv_p, lane, bad);
  }
}

// tile e of the lower 16x16 tiles (row-major over the triangle) of the trailing update of the
// step whose diagonal block starts at j0: A_ij -= L_ip L_jp^T
__device__ __forceinline__ void trailing_tile(double *T, int j0, int e, int lr, int lk) {
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= e) ++ti;
  const int tj = e - ti * (ti + 1) / 2;
  const int ri = j0 + SB + 16 * ti, rj = j0 + SB + 16 * tj;
  v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int kk = 0; kk < SB / 4; ++kk) {
    const int k = j0 + 4 * kk + lk;
    acc = mfma4(T[(ri + lr) * XLD + k], T[(rj + lr) * XLD + k], acc);
  }
#pragma unroll
  for (int g = 0; g < 4; ++g) T[(ri + lk + 4 * g) * XLD + rj + lr] -= acc[g];
}

// columns c0 + wq, c0 + wq + nwq, ... (< c0 + 32) of L back to global: two rows per lane
__device__ __forceinline__ void store_l_columns(const double *T, double *A, int n_pad, int c0, int wq, int nwq,
                                                int lane) {
  const int i2 = lane * 2;
  for (int c = c0 + wq; c < c0 + SB; c += nwq)
    *reinterpret_cast<double2 *>(A + i2 + (int64_t)n_pad * c) =
        make_double2(i2 >= c ? T[i2 * XLD + c] : 0.0, i2 + 1 >= c ? T[(i2 + 1) * XLD + c] : 0.0);
}

// The inverted 32 x 32 diagonal block b of the image (transposed above the diagonal, its diagonal in dinv)
// to the inverse image in global memory, lower triangle only; the work of wave `part` of `nparts`.
__device__ __forceinline__ void store_inv_block(const double *T, const double *dinv, double *Inv, int b, int part,
                                                int nparts, int lane) {
  for (int idx = part * 64 + lane; idx < 16 * SB; idx += nparts * 64) {
    const int rp = idx & 15, c = idx >> 4;               // rows 2 rp, 2 rp + 1 and column c of the block
    const int i = SB * b + 2 * rp, cc = SB * b + c;
    if (c <= 2 * rp + 1)
      *reinterpret_cast<double2 *>(Inv + i + SUP * cc) = make_double2(x_sel(T, dinv, i, cc), x_sel(T, dinv, i + 1, cc));
  }
}

// Off-diagonal blocks of X = L^-1 in the LDS image (L in the lower triangle, the inverted
// 32 x 32 diagonal blocks transposed above it, their diagonals in dinv), for all four waves of the
// workgroup.  Ends with a barrier.
__device__ __forceinline__ void inverse_offdiag(double *T, const double *dinv, int wave, int lr, int lk) {
  // Off-diagonal blocks of X = L^-1 by recursive doubling: for a lower-triangular
  // [[A,0],[B,C]] the inverse is [[A^-1,0],[-C^-1 B A^-1, C^-1]].  Level 1 fills X_10 and
  // X_32 (32x32), level 2 the 64x64 block below the first 64 columns.  All products run on
  // MFMA straight from the image; the inner product M = B A^-1 is parked in the (still
  // unused) destination region of the result, transposed like the result itself.
  {
    // ---- level 1: wave = (pair, column strip of 16) --------------------------------------
    {
      const int pair = wave >> 1, cs = wave & 1;
      const int p0 = 64 * pair, i1 = p0 + SB, c0 = p0 + 16 * cs;
      v4d m0 = {0.0, 0.0, 0.0, 0.0}, m1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < SB / 4; ++kk) {
        if (kk < 4 * cs) continue;                       // X_pp is lower-triangular
        const int k = p0 + 4 * kk + lk;
        const double bv = x_sel(T, dinv, k, c0 + lr);
        m0 = mfma4(T[(i1 + lr) * XLD + k], bv, m0);
        m1 = mfma4(T[(i1 + 16 + lr) * XLD + k], bv, m1);
      }
      double *Mt = &T[(c0 + lr) * XLD + i1];             // M[row][col = lr] at Mt[row]
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Mt[lk + 4 * g] = m0[g];
        Mt[16 + lk + 4 * g] = m1[g];
      }
      double bm[SB / 4];
#pragma unroll
      for (int kk = 0; kk < SB / 4; ++kk) bm[kk] = Mt[4 * kk + lk];
      v4d o0 = {0.0, 0.0, 0.0, 0.0}, o1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < SB / 4; ++kk) {
        const int k = i1 + 4 * kk + lk;
        if (kk < 4) o0 = mfma4(x_sel(T, dinv, i1 + lr, k), bm[kk], o0);
        o1 = mfma4(x_sel(T, dinv, i1 + 16 + lr, k), bm[kk], o1);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        Mt[lk + 4 * g] = -o0[g];
        Mt[16 + lk + 4 * g] = -o1[g];
      }
    }
    __syncthreads();
    
    // ---- level 2, inner product M = L_BA X_AA: wave = 16-row tile of the lower 64 rows -----
    {
      const int ti = wave;
      v4d acc[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) {
        const int k = 4 * kk + lk;
        const double av = T[(64 + 16 * ti + lr) * XLD + k];
#pragma unroll
        for (int cs = 0; cs <= kk / 4; ++cs)             // X_AA[k][c] = 0 for k < c
          acc[cs] = mfma4(av, x_sel(T, dinv, k, 16 * cs + lr), acc[cs]);
      }
#pragma unroll
      for (int cs = 0; cs < 4; ++cs)
#pragma unroll
        for (int g = 0; g < 4; ++g) T[(16 * cs + lr) * XLD + 64 + 16 * ti + lk + 4 * g] = acc[cs][g];
    }
    __syncthreads();
    
    // ---- level 2, X_BA = -X_BB M: wave = column strip of 16 (reads and writes only its own
    // 16 image rows of the parked M) ----------------------------------------------------------
    {
      const int cs = wave;
      double *Mt = &T[(16 * cs + lr) * XLD + 64];
      double bm[16];
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) bm[kk] = Mt[4 * kk + lk];
      v4d o[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) o[q] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < 16; ++kk)
#pragma unroll
        for (int ti = kk / 4; ti < 4; ++ti)              // X_BB is lower-triangular
          o[ti] = mfma4(x_sel(T, dinv, 64 + 16 * ti + lr, 64 + 4 * kk + lk), bm[kk], o[ti]);
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int g = 0; g < 4; ++g) Mt[16 * ti + lk + 4 * g] = -o[ti][g];
    }
    __syncthreads();
  }
}

// One workgroup (4 waves) factors a 128x128 block in LDS and inverts the factor, blocked by
// 32: wave 0 factors and inverts the 32x32 diagonal block in registers, then all four waves
// apply the panel solve and the trailing update with FP64 MFMA straight from LDS.  The
// off-diagonal blocks of the inverse follow by block substitution, again on MFMA.
// OFFDIAG = false leaves the inverse at its 32 x 32 diagonal blocks: enough for the panel solve by
// substitution (trsm_panel_kernel), a quarter of the kernel's time saved on the critical path;
// block_inverse_kernel completes the 128 x 128 inverses later for the callers that multiply by them.
constexpr int DIAG_LDS_DOUBLES = TILE * XLD + TILE + 2 * SB;    // image, inverted diagonal, pivot-column exchange

template <bool OFFDIAG>
__device__ __forceinline__ void potrf_diag_body(double *__restrict__ lds, double *__restrict__ S, int n_pad, int kb,
                                                double *__restrict__ inv_diag, int *__restrict__ info, int z) {
  double *T = lds, *dinv = lds + TILE * XLD, *ucol = dinv + TILE;
  __builtin_amdgcn_s_setprio(3);          // critical path: ahead of any throughput wave on the same SIMD
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double *A = S + (int64_t)n_pad * n_pad * z + (int64_t)kb * TILE * (n_pad + 1);
  double *Inv = inv_diag + inv_group_elems(n_pad) * z + inv_block_offset(kb);   // ld = SUP
  STAMP(0);

  // block in, lower triangle only (an entry above the diagonal is stored as zero without being loaded).
  // First the 32 columns that the factorisation of the first 32 x 32 block and its panel read, by all four
  // waves; the other 96 columns come in beside that factorisation, on the three waves that have nothing
  // else to do then.  All loads of a batch are issued before the first is used.
  const int i2 = (tid & 63) * 2;                        // this thread's rows i2, i2 + 1
  {
    const int cq = tid >> 6;                            // columns cq + 4 s
    double2 v[8];
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const int c = cq + 4 * s2;
      v[s2] = i2 + 1 >= c ? *reinterpret_cast<const double2 *>(A + i2 + (int64_t)n_pad * c) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int s2 = 0; s2 < 8; ++s2) {
      const int c = cq + 4 * s2;
      T[i2 * XLD + c] = (i2 >= c) ? v[s2].x : 0.0;
      T[(i2 + 1) * XLD + c] = (i2 + 1 >= c) ? v[s2].y : 0.0;
    }
  }
  __syncthreads();
  STAMP(1);

  for (int p = 0; p < TILE / SB; ++p) {
    const int i0 = p * SB;
    double *Tp = &T[i0 * XLD + i0];
    if (wave == 0) {
      const int r = lane & 31, half = lane >> 5;
      double x[SB], uc[SB], un[SB];
      // both halves issue the same unconditional loads; the high half then swaps in the identity
#pragma unroll
      for (int c = 0; c < SB; ++c) x[c] = Tp[r * XLD + c];
#pragma unroll
      for (int c = 0; c < SB; ++c) x[c] = half ? (c == r ? 1.0 : 0.0) : x[c];
      ucol[lane] = x[0];
#pragma unroll
      for (int c = 1; c < SB; ++c) uc[c] = ucol[c];
      int bad = 0;
      blk_range<0, SB>(x, uc, un, ucol, &dinv[i0], lane, bad);
      if (bad && lane == 0) atomicMax(info, kb * TILE + i0 + bad);
      // row r of the image: L[r][c] for c <= r from the low half, X[c][r] (transposed) for
      // c > r from the high half.  Every lane stores every element, the lane that does not
      // own it into a dump slot: 32 plain stores instead of 32 divergent branches.
      // (the dump slot is the row's padding column of the image)
      double *Trow = &T[(i0 + r) * XLD];
#pragma unroll
      for (int c = 0; c < SB; ++c) {
        const int own = ((int)(c > r) ^ half) ^ 1;            // low half: c <= r, high half: c > r
        Trow[own ? i0 + c : TILE + half] = x[c];
      }
      if (p == 0) STAMP(23);
    } else if (p == 0) {
      // columns 32..127 of the block: wave w takes columns 32 + (w - 1) + 3 s
      STAMP1(20);
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        double2 v[16];
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const int c = SB + (wave - 1) + 3 * (s2 + 16 * half);
          v[s2] = i2 + 1 >= c ? *reinterpret_cast<const double2 *>(A + i2 + (int64_t)n_pad * c) : make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int s2 = 0; s2 < 16; ++s2) {
          const int c = SB + (wave - 1) + 3 * (s2 + 16 * half);
          T[i2 * XLD + c] = (i2 >= c) ? v[s2].x : 0.0;
          T[(i2 + 1) * XLD + c] = (i2 + 1 >= c) ? v[s2].y : 0.0;
        }
        STAMP1(21 + half);
      }
    } else {
      // beside the factorisation of block p, on the other three SIMDs: what is left of step p-1
      // (its three tiles that make up block p were waited for).  None of it touches block p or reads
      // what wave 0 writes.
      const int j0 = i0 - SB, NTq = (TILE - j0 - SB) / 16;
      for (int e = 3 + (wave - 1); e < NTq * (NTq + 1) / 2; e += 3) trailing_tile(T, j0, e, lr, lk);
      // The finished column blocks of L, and (without the off-diagonal blocks of the inverse, when nothing else
      // of the inverse image is written) the inverted diagonal blocks, leave from here.  Beside block 1 these
      // waves have the most tiles of all steps (18) and already outlast the factorisation: what block 0
      // finished waits for the step after.
      if (p >= 2)
        for (int q = p == 2 ? 0 : 2; q < p; ++q) {
          store_l_columns(T, A, n_pad, q * SB, wave - 1, 3, lane);
          if (!OFFDIAG) store_inv_block(T, dinv, Inv, q, wave - 1, 3, lane);
        }
    }
    __syncthreads();
    STAMP(2 + 3 * p);

    const int R = TILE - i0 - SB;                  // rows below the diagonal block
    const int NT = R / 16;
    // panel: L_ip = A_ip * X_pp^T, a wave owns whole 16-row strips (reads before its writes)
    for (int rb = wave; rb < NT; rb += 4) {
      const int ib = i0 + SB + 16 * rb;
      v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int kk = 0; kk < SB / 4; ++kk) {
        const int k = 4 * kk + lk;
        const double av = T[(ib + lr) * XLD + i0 + k];
        const double b0 = x_at(T, dinv, i0 + lr, i0 + k);
        const double b1 = x_at(T, dinv, i0 + 16 + lr, i0 + k);
        acc0 = mfma4(av, b0, acc0);
        acc1 = mfma4(av, b1, acc1);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        T[(ib + lk + 4 * g) * XLD + i0 + lr] = acc0[g];
        T[(ib + lk + 4 * g) * XLD + i0 + 16 + lr] = acc1[g];
      }
    }
    __syncthreads();
    STAMP(3 + 3 * p);
    // the three tiles of the trailing update that make up the next diagonal block; the other
    // tiles run beside that block's factorisation
    if (NT >= 2 && wave < 3) trailing_tile(T, i0, wave, lr, lk);
    __syncthreads();
    STAMP(4 + 3 * p);
  }

  // the diagonal tiles of the trailing update also wrote their upper halves: L's strict upper
  // triangle inside each 32-block was re-zeroed by the factor step, the rest of the upper
  // triangle is where X goes next.  The last column block of L back to global first.
  store_l_columns(T, A, n_pad, TILE - SB, wave, 4, lane);
  __syncthreads();
  STAMP(14);

  if (OFFDIAG) inverse_offdiag(T, dinv, wave, lr, lk);
  STAMP(17);

  if (OFFDIAG) {
    const int cq = tid >> 6;
#pragma unroll 8
    for (int s2 = 0; s2 < 32; ++s2) {
      const int c = cq + 4 * s2;
      // the inverse image's strict upper triangle is zero from its allocation on and stays so
      if (c <= i2 + 1)
        *reinterpret_cast<double2 *>(Inv + i2 + SUP * c) =
            make_double2(x_sel(T, dinv, i2, c), x_sel(T, dinv, i2 + 1, c));
    }
  } else {
    store_inv_block(T, dinv, Inv, TILE / SB - 1, wave, 4, lane);
  }
  STAMP(18);
}

template <bool OFFDIAG>
__global__ __launch_bounds__(256) void potrf_diag_kernel(double *__restrict__ S, int n_pad, int kb,
                                                        double *__restrict__ inv_diag, int nb,
                                                        int *__restrict__ info) {
  __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS_DOUBLES];
  potrf_diag_body<OFFDIAG>(lds, S, n_pad, kb, inv_diag, info, blockIdx.x);
}

// Panel solve by substitution, L_ik = A_ik L_kk^-T for every block row i > k, from the factor's
// diagonal block and its inverted 32 x 32 diagonal blocks.  One wave owns 16 rows of the panel and
// keeps their solution transposed in MFMA accumulators, 8 tiles of (16 panel columns x 16 rows):
// with X^T = L_kk^-1 A^T in 32-column blocks p,
//   X_p^T = X_pp (A_p^T - sum_{q<p} L_pq X_q^T),
// every product has its left factor in LDS (read as the MFMA's first operand, 16 consecutive
// rows of one column per quarter-wave) and its right factor in accumulators -- and a 16x16x4 FP64
// accumulator register r of a lane is exactly that lane's second operand of k-step r.  The ten
// 32 x 32 blocks every wave needs (six of L_kk below its diagonal, the four inverted diagonal
// blocks) are staged once per workgroup with all loads in flight together: 80 KB, so that the
// workgroup still fits beside a 74 KB workgroup of the GEMM on the same CU.  Element (i, k) of a
// block sits at k*32 + (i ^ 16 (k & 1)): the two quarter-waves of a fragment read then fall on
// disjoint halves of the banks.  144 MFMAs per wave, one barrier.
constexpr int TRSM_BLOCKS = 10;
__device__ __forceinline__ int trsm_slot(int p, int q) { return p * (p - 1) / 2 + q; }   // L_pq, q < p: 0..5
__device__ __forceinline__ int trsm_at(int i, int k) { return k * SB + (i ^ ((k & 1) << 4)); }

__global__ __launch_bounds__(256) void trsm_panel_kernel(double *__restrict__ S, int n_pad, int kb,
                                                         const double *__restrict__ inv_diag) {
  __shared__ __attribute__((aligned(16))) double blk[TRSM_BLOCKS][SB * SB];
  // on the factorisation's critical path, possibly on a SIMD shared with a wave of a throughput
  // launch (the chunks of nu = L*Z on the second stream): issue first
  __builtin_amdgcn_s_setprio(3);
  const int z = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const int row0 = (kb + 1) * TILE + 64 * blockIdx.x + 16 * wave;
  double *Sz = S + (int64_t)n_pad * n_pad * z;
  const double *Lkk = Sz + (int64_t)kb * TILE * (n_pad + 1);             // diagonal block, ld = n_pad
  const double *Inv = inv_diag + inv_group_elems(n_pad) * z + inv_block_offset(kb);   // ld = SUP
  double *P = Sz + row0 + (int64_t)n_pad * kb * TILE;                    // this wave's 16 panel rows
  // stage the blocks: thread t carries rows (2 (t & 15), +1) of columns (t >> 4) and (t >> 4) + 16
  double2 st[TRSM_BLOCKS][2];
  const int si = (tid & 15) << 1, sk = tid >> 4;
#pragma unroll
  for (int p = 1; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < p; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        st[trsm_slot(p, q)][h] = *reinterpret_cast<const double2 *>(Lkk + 32 * p + si + (int64_t)n_pad * (32 * q + sk + 16 * h));
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      st[6 + p][h] = *reinterpret_cast<const double2 *>(Inv + 32 * p + si + (int64_t)SUP * (32 * p + sk + 16 * h));
  v4d xt[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) xt[t][r] = P[lr + (int64_t)n_pad * (16 * t + lk + 4 * r)];
#pragma unroll
  for (int b = 0; b < TRSM_BLOCKS; ++b)
#pragma unroll
    for (int h = 0; h < 2; ++h)      // rows si, si + 1 stay neighbours under the swizzle (it flips bit 4 only)
      *reinterpret_cast<double2 *>(&blk[b][trsm_at(si, sk + 16 * h)]) = st[b][h];
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    v4d w[2] = {xt[2 * p], xt[2 * p + 1]};
#pragma unroll
    for (int q = 0; q < p; ++q) {
      const double *Lpq = blk[trsm_slot(p, q)];
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const double x = xt[2 * q + (ks >> 2)][ks & 3];
        const int k = 4 * ks + lk;
        w[0] = mfma4(-Lpq[trsm_at(lr, k)], x, w[0]);
        w[1] = mfma4(-Lpq[trsm_at(16 + lr, k)], x, w[1]);
      }
    }
    const double *Xpp = blk[6 + p];
    v4d o0 = {0.0, 0.0, 0.0, 0.0}, o1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {      // X_pp is lower-triangular: its upper 16 rows stop at column 16
      const double x = w[ks >> 2][ks & 3];
      const int k = 4 * ks + lk;
      if (ks < 4) o0 = mfma4(Xpp[trsm_at(lr, k)], x, o0);
      o1 = mfma4(Xpp[trsm_at(16 + lr, k)], x, o1);
    }
    xt[2 * p] = o0;
    xt[2 * p + 1] = o1;
  }
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) P[lr + (int64_t)n_pad * (16 * t + lk + 4 * r)] = xt[t][r];
}

int launch_trsm_panel(hipStream_t st, double *S, int n_pad, int kb, int batch, const double *inv_diag) {
  const int rows = n_pad - (kb + 1) * TILE;
  if (rows <= 0) return GPIRT_OK;
  hipLaunchKernelGGL(trsm_panel_kernel, dim3(rows / 64, batch), dim3(256), 0, st, S, n_pad, kb, inv_diag);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------------
// The steps of a 512-column super-panel with the diagonal super-block FIRST.  The chain of small
// kernels that the factorisation's critical path consists of (diagonal block, panel solve, update
// of the rest of the super-panel, once per 128 columns) costs what its launches' first loads and
// stores cost, whatever the number of rows -- so the rows BELOW the diagonal super-block leave the
// chain: per 128-column step the chain only touches the at most 384 rows of the super-block
// (potrf_inpanel_kernel: panel solve and update in ONE launch), and the rows below take each block
// column as a step of their own (tall_step_body: solve, then update of the super-panel's later
// block columns) that rides in the NEXT diagonal block's launch (potrf_diag_tall_kernel), beside
// the one workgroup that factors that block.
//
// Both kinds of step solve 16 rows per wave by the substitution of trsm_panel_kernel, from the
// same ten 32 x 32 blocks of the diagonal block (here the six blocks below the diagonal are staged
// NEGATED, which takes the sign flips out of the loops).
// ---------------------------------------------------------------------------------------------
struct DiagBlocks { double2 st[TRSM_BLOCKS][2]; };

// raw buffer access: a wave-uniform base (scalar registers), one fixed 32-bit offset per lane and a
// scalar offset per access -- no 64-bit address per lane and access
typedef __amdgpu_buffer_rsrc_t prsrc_t;
__device__ __forceinline__ prsrc_t panel_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);
}
__device__ __forceinline__ double pbuf_ld(prsrc_t r, uint32_t lane_off, uint32_t wave_off) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, wave_off, 0);
  double d;
  __builtin_memcpy(&d, &v, 8);
  return d;
}
__device__ __forceinline__ void pbuf_st(prsrc_t r, uint32_t lane_off, uint32_t wave_off, double d) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  v2u v;
  __builtin_memcpy(&v, &d, 8);
  __builtin_amdgcn_raw_buffer_store_b64(v, r, lane_off, wave_off, 0);
}
__device__ __forceinline__ double2 pbuf_ld2(prsrc_t r, uint32_t lane_off, uint32_t wave_off) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, wave_off, 0);
  double2 d;
  __builtin_memcpy(&d, &v, 16);
  return d;
}

// thread t carries rows (2 (t & 15), +1) of columns (t >> 4) and (t >> 4) + 16 of each block
__device__ __forceinline__ void diag_blocks_load(DiagBlocks &d, const double *__restrict__ Lkk, int64_t n_pad,
                                                 const double *__restrict__ Inv, int tid) {
  const int si = (tid & 15) << 1, sk = tid >> 4;
#pragma unroll
  for (int p = 1; p < 4; ++p)
#pragma unroll
    for (int q = 0; q < p; ++q)
#pragma unroll
      for (int h = 0; h < 2; ++h)
        d.st[trsm_slot(p, q)][h] = *reinterpret_cast<const double2 *>(Lkk + 32 * p + si + n_pad * (32 * q + sk + 16 * h));
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      d.st[6 + p][h] = *reinterpret_cast<const double2 *>(Inv + 32 * p + si + (int64_t)SUP * (32 * p + sk + 16 * h));
}

__device__ __forceinline__ void diag_blocks_store(const DiagBlocks &d, double (*blk)[SB * SB], int tid) {
  const int si = (tid & 15) << 1, sk = tid >> 4;
#pragma unroll
  for (int b = 0; b < TRSM_BLOCKS; ++b)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double sg = b < 6 ? -1.0 : 1.0;
      *reinterpret_cast<double2 *>(&blk[b][trsm_at(si, sk + 16 * h)]) = make_double2(sg * d.st[b][h].x, sg * d.st[b][h].y);
    }
}

// xt (16 rows of a panel, transposed in accumulator layout: xt[t][r] of lane (lr, lk) is the row's
// entry in column 16 t + lk + 4 r) <- xt L_kk^-T
__device__ __forceinline__ void solve16(v4d (&xt)[8], const double (*blk)[SB * SB], int lr, int lk) {
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    v4d w[2] = {xt[2 * p], xt[2 * p + 1]};
#pragma unroll
    for (int q = 0; q < p; ++q) {
      const double *Lpq = blk[trsm_slot(p, q)];      // negated
      // (the fences keep the reads of one block together: unfenced, the compiler lifts the reads of all
      // ten blocks to the top, 300 registers' worth)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const double x = xt[2 * q + (ks >> 2)][ks & 3];
        const int k = 4 * ks + lk;
        w[0] = mfma4(Lpq[trsm_at(lr, k)], x, w[0]);
        w[1] = mfma4(Lpq[trsm_at(16 + lr, k)], x, w[1]);
      }
    }
    const double *Xpp = blk[6 + p];
    v4d o0 = {0.0, 0.0, 0.0, 0.0}, o1 = {0.0, 0.0, 0.0, 0.0};
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {      // X_pp is lower-triangular: its upper 16 rows stop at column 16
      const double x = w[ks >> 2][ks & 3];
      const int k = 4 * ks + lk;
      if (ks < 4) o0 = mfma4(Xpp[trsm_at(lr, k)], x, o0);
      o1 = mfma4(Xpp[trsm_at(16 + lr, k)], x, o1);
    }
    xt[2 * p] = o0;
    xt[2 * p + 1] = o1;
  }
  __builtin_amdgcn_sched_barrier(0);
}

// Copy of the raw rows [(kb+1) TILE, oe TILE) of block column kb, [RAW_LD x TILE] per group: what
// potrf_inpanel_kernel reads while its own waves overwrite those rows in place.
constexpr int RAW_LD = 3 * TILE;
constexpr int64_t RAW_ELEMS = (int64_t)RAW_LD * TILE;
static_assert(2 * RAW_ELEMS == POTRF_RAW_ELEMS, "raw panel buffers");

__global__ __launch_bounds__(256) void potrf_rawcopy_kernel(const double *__restrict__ S, int n_pad, int kb, int rows,
                                                            double *__restrict__ raw, int64_t raw_stride) {
  const int z = blockIdx.y, c = blockIdx.x;
  const double *src = S + (int64_t)n_pad * n_pad * z + (int64_t)(kb + 1) * TILE + (int64_t)n_pad * (kb * TILE + c);
  double *dst = raw + raw_stride * z + (int64_t)RAW_LD * c;
  const int r = 2 * threadIdx.x;
  if (r < rows) *reinterpret_cast<double2 *>(dst + r) = *reinterpret_cast<const double2 *>(src + r);
}

// One 128-column step inside the diagonal super-block [o, oe): block column kb is factored; the
// rows [(kb+1) TILE, oe TILE) of it are solved, L_ik = A_ik L_kk^-T, and the block columns kb+1..oe-1
// of those rows take the panel's update, A_ij -= L_ik L_jk^T (lower 16 x 16 tiles), in ONE launch.
// A wave owns one 16 x 16 tile (gi, gj) of the update and solves the two 16-row groups it needs
// itself -- from the raw copy of the panel (raw_in), because the waves of the diagonal tiles write
// the solved rows over the panel in place while other waves still read the raw rows.  The solved
// groups are, as they stand in the accumulators, the two operands of the tile's product.  The tiles
// of block column kb+1 below its diagonal block are the next step's raw panel: they are stored a
// second time into raw_out.
__global__ __launch_bounds__(256) void potrf_inpanel_kernel(double *__restrict__ S, int n_pad, int kb, int oe,
                                                            const double *__restrict__ inv_diag,
                                                            const double *__restrict__ raw_in,
                                                            double *__restrict__ raw_out, int64_t raw_stride) {
  __shared__ __attribute__((aligned(16))) double blk[TRSM_BLOCKS][SB * SB];
  __builtin_amdgcn_s_setprio(3);
  const int z = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double *Sz = S + (int64_t)n_pad * n_pad * z;
  const double *Lkk = Sz + (int64_t)kb * TILE * (n_pad + 1);
  const double *Inv = inv_diag + inv_group_elems(n_pad) * z + inv_block_offset(kb);
  const double *Rin = raw_in + raw_stride * z;
  double *Rout = raw_out + raw_stride * z;
  const int G = (oe - kb - 1) * (TILE / 16), ntiles = G * (G + 1) / 2;
  const int T = blockIdx.x * 4 + wave;
  const bool live = T < ntiles;                       // uniform over the wave
  int gi = 0, gj = 0;
  if (live) {
    gi = (int)((sqrtf(8.0f * (float)T + 1.0f) - 1.0f) * 0.5f);
    while ((gi + 1) * (gi + 2) / 2 <= T) ++gi;
    while (gi * (gi + 1) / 2 > T) --gi;
    gj = T - gi * (gi + 1) / 2;
  }
  DiagBlocks d;
  diag_blocks_load(d, Lkk, n_pad, Inv, tid);
  v4d xi[8], xj[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      xi[t][r] = Rin[(16 * gi + lr) + (int64_t)RAW_LD * (16 * t + lk + 4 * r)];
      xj[t][r] = Rin[(16 * gj + lr) + (int64_t)RAW_LD * (16 * t + lk + 4 * r)];
    }
  diag_blocks_store(d, blk, tid);
  __syncthreads();
  if (!live) return;
  const int rows0 = (kb + 1) * TILE;
  // the tile of the update comes in early: its latency hides under the solves
  double *Ct = Sz + (rows0 + 16 * gi + lr) + (int64_t)n_pad * (rows0 + 16 * gj + lk);
  v4d c;
#pragma unroll
  for (int r = 0; r < 4; ++r) c[r] = Ct[(int64_t)n_pad * (4 * r)];
  solve16(xi, blk, lr, lk);
  if (gi != gj) {
    solve16(xj, blk, lr, lk);
  } else {
#pragma unroll
    for (int t = 0; t < 8; ++t) xj[t] = xi[t];
    // this wave owns the row group: the solved rows go over the panel
    double *P = Sz + (rows0 + 16 * gi + lr) + (int64_t)n_pad * (kb * TILE + lk);
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) P[(int64_t)n_pad * (16 * t + 4 * r)] = xi[t][r];
  }
  // acc[r] of lane (lr, lk) = sum_c X_j[4 r + lk][c] X_i[lr][c]; two chains of 16
  v4d a0 = {0.0, 0.0, 0.0, 0.0}, a1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int t = 0; t < 8; t += 2)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      a0 = mfma4(xj[t][r], xi[t][r], a0);
      a1 = mfma4(xj[t + 1][r], xi[t + 1][r], a1);
    }
  const bool next_panel = gj < TILE / 16 && gi >= TILE / 16;    // block column kb+1, below its diagonal block
  double *Rt = Rout + (16 * gi - TILE + lr) + (int64_t)RAW_LD * (16 * gj + lk);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const double v = c[r] - (a0[r] + a1[r]);
    Ct[(int64_t)n_pad * (4 * r)] = v;
    if (next_panel) Rt[(int64_t)RAW_LD * (4 * r)] = v;
  }
}

// pieces of a factor block staged through LDS: 128 rows x 32 columns, element (i, k) at piece_at(i, k)
constexpr int TALL_PIECE = 32;
__device__ __forceinline__ int piece_at(int i, int k) { return k * TILE + (i ^ ((k & 1) << 4)); }

// One block column of the same chain as its own step, right-looking: the 64 rows of workgroup `wg`
// below the diagonal super-block [o, o + nc) solve block column o + c, X_c = A_c L_cc^-T, and update the
// block columns after it, A_c'' -= X_c L_c''c^T (c'' > c), whose blocks L_c''c the step inside the
// super-block has just written.  The step needs the factored diagonal block o + c and nothing of the
// later steps: it runs beside the NEXT diagonal block's kernel, in the same launch
// (potrf_diag_tall_kernel), so the chain does not wait for it -- and the heaviest step is the first,
// the lightest (the solve alone) the one that is left over behind the last diagonal block.
__device__ __forceinline__ void tall_step_body(double *__restrict__ lds, double *__restrict__ S, int n_pad, int o, int nc,
                                               int c, const double *__restrict__ inv_diag, int z, int wg) {
  double (*blk)[SB * SB] = reinterpret_cast<double (*)[SB * SB]>(lds);
  double *piece = lds + TRSM_BLOCKS * SB * SB;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  double *Sz = S + (int64_t)n_pad * n_pad * z;
  const int row0 = (o + nc) * TILE + 64 * wg + 16 * wave;
  const int pi = (tid & 63) << 1, pk = tid >> 6;
  const uint32_t colb = (uint32_t)n_pad * 8u;
  const uint32_t piece_off = (uint32_t)pi * 8u + colb * (uint32_t)pk, row_off = (uint32_t)lr * 8u + colb * (uint32_t)lk;
  // piece q of L[block row o + cc, block column o + c]: 128 rows x 32 columns
  auto piece_load = [&](double2 (&pr)[8], int cc, int q) {
    const prsrc_t rs = panel_rsrc(Sz + (int64_t)(o + cc) * TILE + (int64_t)n_pad * ((o + c) * TILE + TALL_PIECE * q));
#pragma unroll
    for (int h = 0; h < 8; ++h) pr[h] = pbuf_ld2(rs, piece_off, colb * (uint32_t)(4 * h));
  };
  const prsrc_t P = panel_rsrc(Sz + row0 + (int64_t)n_pad * ((o + c) * TILE));
  v4d x[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) x[t][r] = pbuf_ld(P, row_off, colb * (uint32_t)(16 * t + 4 * r));
  {
    DiagBlocks d;
    diag_blocks_load(d, Sz + (int64_t)(o + c) * TILE * (n_pad + 1), n_pad,
                     inv_diag + inv_group_elems(n_pad) * z + inv_block_offset(o + c), tid);
    diag_blocks_store(d, blk, tid);
  }
  double2 pr[8];
  if (c + 1 < nc) piece_load(pr, c + 1, 0);
  __syncthreads();
  solve16(x, blk, lr, lk);
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) pbuf_st(P, row_off, colb * (uint32_t)(16 * t + 4 * r), x[t][r]);
  for (int cc = c + 1; cc < nc; ++cc) {
    const prsrc_t Q = panel_rsrc(Sz + row0 + (int64_t)n_pad * ((o + cc) * TILE));
    v4d w[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) w[t][r] = pbuf_ld(Q, row_off, colb * (uint32_t)(16 * t + 4 * r));
#pragma unroll
    for (int q = 0; q < TILE / TALL_PIECE; ++q) {
      __syncthreads();
#pragma unroll
      for (int h = 0; h < 8; ++h)
        *reinterpret_cast<double2 *>(&piece[piece_at(pi, pk + 4 * h)]) = make_double2(-pr[h].x, -pr[h].y);
      __syncthreads();
      const bool last = cc == nc - 1 && q == TILE / TALL_PIECE - 1;
      if (!last) piece_load(pr, q + 1 < TILE / TALL_PIECE ? cc : cc + 1, q + 1 < TILE / TALL_PIECE ? q + 1 : 0);
      // fragments of k-step ks + 1 are read while the 8 MFMAs of k-step ks issue
      double fa[2][8];
#pragma unroll
      for (int t = 0; t < 8; ++t) fa[0][t] = piece[piece_at(16 * t + lr, lk)];
#pragma unroll
      for (int ks = 0; ks < TALL_PIECE / 4; ++ks) {
        if (ks + 1 < TALL_PIECE / 4) {
#pragma unroll
          for (int t = 0; t < 8; ++t) fa[(ks + 1) & 1][t] = piece[piece_at(16 * t + lr, 4 * (ks + 1) + lk)];
        }
        __builtin_amdgcn_sched_barrier(0);
        const double xv = x[(TALL_PIECE / 16) * q + (ks >> 2)][ks & 3];
#pragma unroll
        for (int t = 0; t < 8; ++t) w[t] = mfma4(fa[ks & 1][t], xv, w[t]);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) pbuf_st(Q, row_off, colb * (uint32_t)(16 * t + 4 * r), w[t][r]);
  }
}

constexpr int TALL_LDS_DOUBLES = TRSM_BLOCKS * SB * SB + TILE * TALL_PIECE;

// workgroup 0 of a group: the diagonal block kb; the others: step c of the rows below the super-block
__global__ __launch_bounds__(256) void potrf_diag_tall_kernel(double *__restrict__ S, int n_pad, int kb,
                                                             double *__restrict__ inv_diag, int *__restrict__ info,
                                                             int o, int nc, int c) {
  static_assert(TALL_LDS_DOUBLES <= DIAG_LDS_DOUBLES, "one buffer serves both roles");
  __shared__ __attribute__((aligned(16))) double lds[DIAG_LDS_DOUBLES];
  if (blockIdx.x == 0) potrf_diag_body<false>(lds, S, n_pad, kb, inv_diag, info, blockIdx.y);
  else { __builtin_amdgcn_s_setprio(2); tall_step_body(lds, S, n_pad, o, nc, c, inv_diag, blockIdx.y, blockIdx.x - 1); }
}

__global__ __launch_bounds__(256) void potrf_tall_step_kernel(double *__restrict__ S, int n_pad, int o, int nc, int c,
                                                             const double *__restrict__ inv_diag) {
  __shared__ __attribute__((aligned(16))) double lds[TALL_LDS_DOUBLES];
  __builtin_amdgcn_s_setprio(3);
  tall_step_body(lds, S, n_pad, o, nc, c, inv_diag, blockIdx.y, blockIdx.x);
}

// diagonal block kb beside step c of the rows below the super-block [o, oe) (c < 0: none)
int launch_potrf_diag_tall(hipStream_t st, double *S, int n_pad, int kb, int batch, double *inv_diag, int *d_info,
                           int o, int oe, int c) {
  const int rows = n_pad - oe * TILE;
  const int tall = c >= 0 && rows > 0 ? rows / 64 : 0;
  hipLaunchKernelGGL(potrf_diag_tall_kernel, dim3(1 + tall, batch), dim3(256), 0, st, S, n_pad, kb, inv_diag, d_info,
                     o, oe - o, c);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

int launch_potrf_tall_step(hipStream_t st, double *S, int n_pad, int o, int oe, int c, int batch, const double *inv_diag) {
  const int rows = n_pad - oe * TILE;
  if (rows <= 0) return GPIRT_OK;
  hipLaunchKernelGGL(potrf_tall_step_kernel, dim3(rows / 64, batch), dim3(256), 0, st, S, n_pad, o, oe - o, c, inv_diag);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

int launch_potrf_rawcopy(hipStream_t st, const double *S, int n_pad, int kb, int oe, int batch, double *raw) {
  const int rows = (oe - kb - 1) * TILE;
  if (rows <= 0) return GPIRT_OK;
  hipLaunchKernelGGL(potrf_rawcopy_kernel, dim3(TILE, batch), dim3(256), 0, st, S, n_pad, kb, rows, raw, 2 * RAW_ELEMS);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// raw: two [RAW_LD x TILE] buffers per group (2 RAW_ELEMS doubles each group); step kb of the super-panel
// that begins at block column o reads buffer (kb - o) & 1 and leaves the next step's panel in the other
int launch_potrf_inpanel(hipStream_t st, double *S, int n_pad, int kb, int o, int oe, int batch,
                         const double *inv_diag, double *raw) {
  const int G = (oe - kb - 1) * (TILE / 16);
  if (G <= 0) return GPIRT_OK;
  const int ntiles = G * (G + 1) / 2, par = (kb - o) & 1;
  hipLaunchKernelGGL(potrf_inpanel_kernel, dim3((ntiles + 3) / 4, batch), dim3(256), 0, st, S, n_pad, kb, oe, inv_diag,
                     raw + par * RAW_ELEMS, raw + (par ^ 1) * RAW_ELEMS, 2 * RAW_ELEMS);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// Completes the inverse of every 128 x 128 diagonal block from its inverted 32 x 32 diagonal
// blocks (what potrf_diag_kernel<false> leaves): one workgroup per block rebuilds the LDS image
// (L below, the 32-blocks' inverses transposed above) and runs the same block substitution.
__global__ __launch_bounds__(256) void block_inverse_kernel(const double *__restrict__ S, int n_pad,
                                                           double *__restrict__ inv_diag) {
  __shared__ __attribute__((aligned(16))) double T[TILE * XLD];
  __shared__ __attribute__((aligned(16))) double dinv[TILE];
  const int kb = blockIdx.x, z = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lr = lane & 15, lk = lane >> 4;
  const double *A = S + (int64_t)n_pad * n_pad * z + (int64_t)kb * TILE * (n_pad + 1);
  double *Inv = inv_diag + inv_group_elems(n_pad) * z + inv_block_offset(kb);
  const int i2 = (tid & 63) * 2, cq = tid >> 6;
  for (int s2 = 0; s2 < 32; ++s2) {
    const int c = cq + 4 * s2;
    const double2 l = *reinterpret_cast<const double2 *>(A + i2 + (int64_t)n_pad * c);
    const double2 x = *reinterpret_cast<const double2 *>(Inv + i2 + (int64_t)SUP * c);
    const bool same = (c & ~(SB - 1)) == (i2 & ~(SB - 1));       // inside a diagonal 32-block
    // L strictly below the diagonal stays where it is; the 32-block's inverse goes to the
    // transposed position above the diagonal, its diagonal to dinv
    if (i2 >= c) T[i2 * XLD + c] = l.x;
    if (i2 + 1 >= c) T[(i2 + 1) * XLD + c] = l.y;
    if (same) {
      if (i2 > c) T[c * XLD + i2] = x.x; else if (i2 == c) dinv[c] = x.x;
      if (i2 + 1 > c) T[c * XLD + i2 + 1] = x.y; else if (i2 + 1 == c) dinv[c] = x.y;
    }
  }
  __syncthreads();
  inverse_offdiag(T, dinv, wave, lr, lk);
#pragma unroll 8
  for (int s2 = 0; s2 < 32; ++s2) {
    const int c = cq + 4 * s2;
    if (c <= i2 + 1)
      *reinterpret_cast<double2 *>(Inv + i2 + SUP * c) =
          make_double2(x_sel(T, dinv, i2, c), x_sel(T, dinv, i2 + 1, c));
  }
}

int launch_block_inverses(hipStream_t st, const double *S, int n_pad, int batch, double *inv_diag) {
  hipLaunchKernelGGL(block_inverse_kernel, dim3(n_pad / TILE, batch), dim3(256), 0, st, S, n_pad, inv_diag);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

int launch_potrf_diag(hipStream_t st, double *S, int n_pad, int kb, int batch, double *inv_diag, int nb,
                      int *d_info, bool offdiag) {
  if (offdiag)
    hipLaunchKernelGGL(potrf_diag_kernel<true>, dim3(batch), dim3(256), 0, st, S, n_pad, kb, inv_diag, nb, d_info);
  else
    hipLaunchKernelGGL(potrf_diag_kernel<false>, dim3(batch), dim3(256), 0, st, S, n_pad, kb, inv_diag, nb, d_info);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

}  // namespace gpirt
