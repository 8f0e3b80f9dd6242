// Shared declarations of the gfx950 GPIRT library: error plumbing, the counter-based
// generator, the ordinal-probit cell, padding rules and the kernel launch prototypes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/gpirt.h"

#define GPIRT_LOG_QUAL static __device__ const
#include "log_table.h"

namespace gpirt {

// ---- padding ---------------------------------------------------------------------
// Every dense operand is padded to multiples of TILE in both dimensions so that no MFMA
// tile needs a bounds check.  Pad rows/columns of data matrices are zero; the pad block of
// a kernel matrix is the identity, which keeps its Cholesky factor block-diagonal.
constexpr int TILE = 128;
inline int64_t pad_up(int64_t v, int64_t t = TILE) { return (v + t - 1) / t * t; }

// ---- error plumbing ----------------------------------------------------------------
void set_error(const char *fmt, ...);
#define GP_HIP(expr)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (expr);                                                            \
    if (e_ != hipSuccess) {                                                            \
      gpirt::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, \
                       __LINE__);                                                      \
      return GPIRT_ERR_HIP;                                                            \
    }                                                                                  \
  } while (0)
#define GP_TRY(expr)            \
  do {                          \
    int rc_ = (expr);           \
    if (rc_ != GPIRT_OK) return rc_; \
  } while (0)

// ---- stream tags of the generator (shared with the CPU oracle) -----------------------
enum : uint32_t {
  TAG_F_Z = 1, TAG_F_U = 2, TAG_P_Z = 3, TAG_P_U = 4, TAG_T_Z = 5, TAG_T_U = 6, TAG_INIT = 7
};

// ---- Philox4x32-10, counter = (index, unit, sweep, tag), key = seed -------------------
__host__ __device__ inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    uint32_t n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__host__ __device__ inline double u53(uint32_t hi, uint32_t lo) {
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6) + 0.5) *
         (1.0 / 9007199254740992.0);
}

__host__ __device__ inline void rng_uniform_pair(uint64_t seed, uint32_t tag, uint32_t sweep,
                                                 uint32_t unit, uint32_t q, double &u0,
                                                 double &u1) {
  uint32_t c[4] = {q, unit, sweep, tag};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  u0 = u53(c[0], c[1]);
  u1 = u53(c[2], c[3]);
}

__host__ __device__ inline double rng_uniform_at(uint64_t seed, uint32_t tag, uint32_t sweep,
                                                 uint32_t unit, uint32_t d) {
  double a, b;
  rng_uniform_pair(seed, tag, sweep, unit, d >> 1, a, b);
  return (d & 1) ? b : a;
}

constexpr double kTwoPi = 6.283185307179586476925286766559;

// Natural logarithm of a positive, normal double for the device, from a 258-row table (log_table.h,
// tools/gen_log_table.py): x = 2^e m with m in [sqrt(1/2), sqrt(2)), the mantissa of m rounded to 7 bits picks the
// nearest centre c (a power of two sits exactly on one: its logarithm is e ln2, correctly rounded, and a sum of
// logs of nothing but ones is exactly zero), and with ic = fl(1/c)
//     log x = e ln2 + log1p(r) - log(ic),   r = fma(m, ic, -1),  |r| <= 2^-8,
// where -log(ic) is tabulated for the ROUNDED ic (the reciprocal's error never enters) and log1p(r) is its Taylor
// polynomial to r^6.  About 25 instructions, no division and no branch, where the library's log takes ~70;
// absolute error below 2^-52 max(1, |log x|).  The sums of log-probabilities (log_like_fn.m:3-4) and the normal
// draws (sample_fs.m:2: Box-Muller's sqrt(-2 log u)) go through it.
constexpr int LOG_LDS_DOUBLES = 516;

// every thread of the workgroup calls this once, followed by __syncthreads()
__device__ inline void log_table_to_lds(double *s) {
  for (int e = threadIdx.x; e < LOG_LDS_DOUBLES; e += blockDim.x) s[e] = kLogTab[e];
}

// t: the table, in LDS (log_table_to_lds) or kLogTab itself
__device__ inline double log_pos(double x, const double *t) {
  const int hi = __double2hiint(x);
  const int mant = hi & 0xFFFFF;
  const int low = mant >= GPIRT_LOG_SQRT2_HI ? 1 : 0;                 // mantissas from sqrt(2) on: the binade below
  const int e = (hi >> 20) - 1023 + low;
  const double m = __hiloint2double(mant | (low ? 0x3FE00000 : 0x3FF00000), __double2loint(x));
  const double2 c = reinterpret_cast<const double2 *>(t)[((mant + 0x1000) >> 13) + 129 * low];
  const double r = fma(m, c.x, -1.0);
  double q = fma(r, -1.0 / 6.0, 0.2);
  q = fma(q, r, -0.25);
  q = fma(q, r, 1.0 / 3.0);
  q = fma(q, r, -0.5);
  const double p = fma(q, r * r, r);
  const double ef = (double)e;
  return fma(ef, GPIRT_LOG_LN2_HI, c.y) + fma(ef, GPIRT_LOG_LN2_LO, p);   // e ln2_hi is exact (42-bit constant)
}

// Box-Muller on one Philox block: two standard normals.  On the device the logarithm is log_pos and the angle
// 2 pi b goes through sincospi(2 b), whose argument is exact and needs no range reduction: the draws of a sweep
// (one pair per two entries of Z) take 113 us where they took 147 with log and sincos(2 pi b).
__host__ __device__ inline void rng_normal_pair(uint64_t seed, uint32_t tag, uint32_t sweep,
                                                uint32_t unit, uint32_t q, double &z0,
                                                double &z1) {
  double a, b;
  rng_uniform_pair(seed, tag, sweep, unit, q, a, b);
  double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = sqrt(-2.0 * log_pos(a, kLogTab));
  sincospi(2.0 * b, &s, &c);
#else
  const double r = sqrt(-2.0 * log(a));
  sincos(kTwoPi * b, &s, &c);
#endif
  z0 = r * c;
  z1 = r * s;
}

__host__ __device__ inline double rng_normal_at(uint64_t seed, uint32_t tag, uint32_t sweep,
                                                uint32_t unit, uint32_t d) {
  double a, b;
  rng_normal_pair(seed, tag, sweep, unit, d >> 1, a, b);
  return (d & 1) ? b : a;
}

// ---- the ordinal-probit cell ----------------------------------------------------------
// log( Phi(zhi) - Phi(zlo) ) for zlo < zhi, on the upper tails when both are positive
// (log_like_fn.m:3-4 with the probit link of sigmoid_fn.m:4, ordinal form
// gpirt-sythetic.R:159-167).  The floor keeps a vanished cell finite.
constexpr double kInvSqrt2 = 0.70710678118654752440084436210485;
constexpr double kPFloor = 1e-300;

__host__ __device__ inline double ordinal_cell_z(double zlo, double zhi) {
  double p;
  if (zlo > 0.0)
    p = 0.5 * erfc(zlo * kInvSqrt2) - 0.5 * erfc(zhi * kInvSqrt2);
  else
    p = 0.5 * erfc(-zhi * kInvSqrt2) - 0.5 * erfc(-zlo * kInvSqrt2);
  if (!(p > kPFloor)) p = kPFloor;
  return log(p);
}

// ---- wave-level sum (64 lanes) ----------------------------------------------------------
__device__ inline double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---- dense launchers (dense.hip) ---------------------------------------------------------
struct GemmArgs {
  const double *A; int64_t lda, strideA;
  const double *B; int64_t ldb, strideB;
  double *C;       int64_t ldc, strideC;
  int M, N, K;       // multiples of TILE, TILE, 16
  int batch;
  double alpha, beta;
  int lower_only;    // compute only tiles with row-block >= col-block (C square)
  int k_trim;        // 1: A is lower-triangular => K range of row-block bi is [0,(bi+1)*TILE)
                     // 2: A^T is used with A lower-triangular => K range is [bi*TILE, K)
                     // 3: as 1, instantiated under its own kernel name (the nu = L*Z launch),
                     //    tiles in the XCD-aware order; 4: as 1 under a third kernel name (the
                     //    rows of nu left over after the factorisation)
  int k_shift;       // k_trim 1 or 3 with the batch index used to split K: slab z holds the K
                     // columns from z*k_shift on, so its trimmed range ends z*k_shift earlier
  int k_base;        // k_trim 1 or 3 on a row range of the triangular matrix that starts at row
                     // k_base (A and C point at that row): the trimmed range ends k_base later
  int lone;          // lower_only or k_trim 1, batch 1: run as a fixed grid of fewer workgroups than
                     // CUs, one to a CU, so that another stream's small kernels find room beside it
};
// C = alpha * op(A) * op(B) + beta * C.
//   AT = false: A(i,k) at A[i + lda*k];  AT = true: A(i,k) at A[k + lda*i]
//   BT = false: B(k,j) at B[k + ldb*j];  BT = true: B(k,j) at B[j + ldb*k]
int launch_gemm(hipStream_t st, bool AT, bool BT, const GemmArgs &a);

// S[p] = K(x_p, x_p) + jitter I on the leading n x n block, identity on the pad block;
// strict upper triangle zero unless `full`.
int launch_k_build(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad,
                   int batch, double ls, double sf, double jitter, double *S, bool full);
// The factor's diagonal blocks are kept inverted, in super-blocks of SUP rows: super-block c
// of group z is a dense lower-triangular SUP x SUP matrix (ld = SUP) at
// inv + (c + nsup*z) * SUP*SUP.  The 128 x 128 diagonal blocks come out of the Cholesky's
// diagonal-block kernel; the off-diagonal blocks are filled in by block substitution
// afterwards, so that the triangular solves advance SUP rows per step.
constexpr int SUP = 4 * TILE;
__host__ __device__ inline int inv_nsup(int n_pad) { return (n_pad / TILE + 3) / 4; }
__host__ __device__ inline int64_t inv_group_elems(int n_pad) { return (int64_t)inv_nsup(n_pad) * SUP * SUP; }
__host__ __device__ inline int64_t inv_block_offset(int kb) {   // 128-block kb inside its group's inverse image
  return (int64_t)(kb / 4) * SUP * SUP + (int64_t)(kb % 4) * TILE * (SUP + 1);
}
// scratch the factorisation needs: 256*256 doubles per super-block and group for the inverted super-blocks,
// and two raw copies of a super-panel's rows inside the diagonal super-block (3 TILE x TILE each) per group
// for the steps inside a super-panel (the two uses never overlap in time)
constexpr int64_t POTRF_RAW_ELEMS = 2 * 3 * (int64_t)TILE * TILE;
inline int64_t potrf_work_elems(int n_pad, int batch) {
  const int64_t inv = (int64_t)inv_nsup(n_pad) * 256 * 256;
  return (inv > POTRF_RAW_ELEMS ? inv : POTRF_RAW_ELEMS) * batch;
}
// In-place lower Cholesky of `batch` padded matrices + the inverted diagonal super-blocks
// (inv must be zero-initialised once); *d_info (device int) is set non-zero when a pivot is
// not positive.
// `hook` (optional) is called on the host right after the launches that finish block columns
// [o, oe) and the trailing update behind them have been enqueued: from then on (in stream order)
// rows [o*TILE, oe*TILE) of L are final and the chain of small kernels of the next block columns begins.
// `timer` (optional) is called right before (begin = 1) and right after (begin = 0) the launch of
// each rank-512 trailing update, with that launch's algorithmic flop count: the caller may bracket
// the launch with events of its own.
// `side` (optional, batch 1): a second stream for the look-ahead schedule.  The rank-512 update
// behind super-panel s is split: its first four block columns (all that the chain of small
// kernels of super-panel s+1 needs) stay on `st`; the rest runs on `side` as a fixed grid of
// fewer workgroups than CUs, beside that chain.  `ev` must hold 2 * ceil(nb / 4) events.  The
// arithmetic per tile is that of the plain schedule: same factor bit for bit.  In stream order
// on `st` the factor is still final when the call returns.  With `side`, `fn` is called when block
// columns [o, oe) are final on `st` and the side stream's share of the update behind them is
// enqueued: what `fn` puts on `side` runs behind that share; `timer` brackets the side stream's
// launches (its `stream` argument says where to record).
struct PotrfHook {
  int (*fn)(void *ctx, int o, int oe); void *ctx;
  int (*timer)(void *ctx, int begin, double flop, hipStream_t stream);
  hipStream_t side;
  hipEvent_t *ev;
};
// mode bits:
//   POTRF_DEFER_INVERSES  the inverted diagonal (super-)blocks, which only callers that multiply by
//                         them read (launch_potrs, the generator form), are left to a later
//                         launch_super_inverses call (work may then be null);
//   POTRF_SUBSTITUTION    the panels below a diagonal block are solved by block substitution from
//                         the block's inverted 32 x 32 diagonal blocks instead of a product with its
//                         whole inverse, which the diagonal-block kernel then does not build (a
//                         quarter of its time, on the factorisation's critical path).
enum { POTRF_DEFER_INVERSES = 1, POTRF_SUBSTITUTION = 2 };
int launch_potrf(hipStream_t st, double *S, int n_pad, int batch, double *inv, int *d_info,
                 double *work, const PotrfHook *hook = nullptr, int mode = 0);
// complete_blocks: the factorisation ran with POTRF_SUBSTITUTION (32 x 32 inverses only so far)
int launch_super_inverses(hipStream_t st, const double *L, int n_pad, int batch, double *inv, double *work,
                          bool complete_blocks = false);
// B <- S^-1 B (two blocked triangular solves) for `batch` systems sharing the layout;
// xtmp holds 4 * SUP * nrhs_pad doubles per system (split-K slabs of the diagonal steps).
int launch_potrs(hipStream_t st, const double *L, const double *inv, int n_pad, int batch,
                 double *B, int64_t ldb, int64_t strideB, int nrhs_pad, double *xtmp);
// the same cross-covariance at an arbitrary list of R nodes (device array), R_pad columns written
int launch_kstar_nodes(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                       double ls, double sf, const double *nodes, int R, int R_pad, double *Kc,
                       int64_t group_stride);
// Pm[r + n_pad*j] = l_j(x_r): barycentric Lagrange basis of the R nodes (weights bw) at the
// inputs; *outside |= 1 when an input lies outside [lo, hi]
int launch_bary(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                const double *nodes, const double *bw, int R, double lo, double hi, double *Pm,
                int64_t group_stride, int *outside);
// A[(row0+i) + ld*j] = v * (i == j), i, j < TILE
int launch_scaled_identity(hipStream_t st, double *A, int64_t ld, int row0, int batch, int64_t stride,
                           double v);
// J[b] = s2 I + slabs[0] + ... + slabs[b-1], TILE x TILE blocks (b < nblk)
int launch_prefix_gram(hipStream_t st, const double *slabs, int nblk, double s2, double *J);
// zero above the diagonal of `batch` TILE x TILE blocks
int launch_tril(hipStream_t st, double *A, int batch);
// in place: block b <- sum of the blocks before it (blocks of `elems` doubles, elems even)
int launch_prefix_blocks(hipStream_t st, double *V, int64_t elems, int nblk);
// out[r + ldo*j] = sum_q x[q*slab_stride + r + ldx*j], rows even (split-K partial results)
int launch_sum_slabs(hipStream_t st, const double *x, int nslab, int64_t slab_stride, int rows,
                     int cols, int64_t ldx, double *out, int64_t ldo);
// Kst[r + n_pad*g] = sf^2 exp(-(xs_g - x_r)^2 / (2 ls^2)), zero outside r<n, g<G.
int launch_kstar_t(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                   double ls, double sf, int G, int G_pad, double lo, double step, double *Kst);

}  // namespace gpirt
