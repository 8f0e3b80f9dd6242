// Dense FP64 pieces of the sweep for gfx950: kernel-matrix build, blocked Cholesky,
// triangular products/solves -- all on one MFMA GEMM core (v_mfma_f64_16x16x4_f64).
//
//   K(theta,theta)+jitter  main.m:29,67 (covSEiso, utils/model.m:4,6); getprob_gpirt.R:41-51
//   chol                   utils/sample_fs.m:2 (inside mvnrnd); generate_data.R:56
//   nu = L z               utils/elliptical_slice.m:64
//   S^-1 B                 the solve inside gp(...), main.m:57-58
#include "common.h"
#include <algorithm>
#include <cstdlib>

namespace gpirt {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int BK = 16;        // k-slab depth held in LDS
// super-tile of the XCD-aware tile order (MODE 4): rows x columns of 128 x 128 tiles that run together on one XCD
#ifndef GPIRT_NU_ST_R
#define GPIRT_NU_ST_R 8
#endif
#ifndef GPIRT_NU_ST_C
#define GPIRT_NU_ST_C 4
#endif
constexpr int NU_ST_R = GPIRT_NU_ST_R, NU_ST_C = GPIRT_NU_ST_C;
constexpr int SMALL_GEMM_TILES = 256;   // at most this many 128x128 tiles go to the 64x64-tile kernel
constexpr int LDS_LD = 144;   // 128 + 16: the four k-rows one ds_read_b64 touches land on
                              // disjoint bank halves

// ---------------------------------------------------------------------------------------
// GEMM core.  One 256-thread workgroup owns a 128x128 tile of C; its four waves own 64x64
// quadrants as 4x4 MFMA tiles of 16x16.  The MFMA is fed transposed (its "A" operand is
// the B-tile, its "B" operand the A-tile) so that the 16 lanes sharing an accumulator
// register hold 16 consecutive rows i of one column j: the epilogue then writes 128-byte
// runs of the column-major C instead of 8-byte scatters.
// ---------------------------------------------------------------------------------------
// Global -> register staging of one 128 x BK operand slab (4 x 16 B per thread) and its
// store into the k-major LDS image.  KCONTIG says which index is contiguous in memory.
// The four registers are separate scalars on purpose: as an array they ended up in scratch.
template <bool KCONTIG>
__device__ __forceinline__ double2 slab_ld(const double *__restrict__ base, int64_t ld, int row0,
                                           int k0, int tid, int p) {
  if (!KCONTIG)
    return *reinterpret_cast<const double2 *>(base + (row0 + ((tid & 63) << 1)) +
                                              ld * (int64_t)(k0 + (tid >> 6) + 4 * p));
  return *reinterpret_cast<const double2 *>(base + (k0 + ((tid & 7) << 1)) +
                                            ld * (int64_t)(row0 + (tid >> 3) + 32 * p));
}

// LDS image of one operand slab.  An operand whose tile index is contiguous in memory is kept
// k-major, [k][index] with row stride LDS_LD; an operand whose K index is contiguous is kept
// index-major, [index][k] with row stride KLD, so that its staging stores stay 16-byte vector
// stores instead of transposing scalar stores (which collided 8-way on the banks).  KLD = 20:
// the 16 index rows of one fragment read start 40 dwords apart, which tiles the 64 banks
// exactly twice.
constexpr int KLD = 20;
template <bool KCONTIG> struct SlabImage { static constexpr int SIZE = KCONTIG ? TILE * KLD : BK * LDS_LD; };

template <bool KCONTIG>
__device__ __forceinline__ void slab_st(double *s, int tid, int p, double2 v) {
  if (!KCONTIG)
    *reinterpret_cast<double2 *>(&s[((tid >> 6) + 4 * p) * LDS_LD + ((tid & 63) << 1)]) = v;
  else
    *reinterpret_cast<double2 *>(&s[((tid >> 3) + 32 * p) * KLD + ((tid & 7) << 1)]) = v;
}

// element (k, idx) of a slab image
template <bool KCONTIG>
__device__ __forceinline__ double slab_at(const double *s, int k, int idx) {
  return KCONTIG ? s[idx * KLD + k] : s[k * LDS_LD + idx];
}

#define TILE_LOAD(KC, base, ld, row0, k0, r)          \
  r##0 = slab_ld<KC>(base, ld, row0, k0, tid, 0);     \
  r##1 = slab_ld<KC>(base, ld, row0, k0, tid, 1);     \
  r##2 = slab_ld<KC>(base, ld, row0, k0, tid, 2);     \
  r##3 = slab_ld<KC>(base, ld, row0, k0, tid, 3)
#define TILE_STORE(KC, s, r)   \
  slab_st<KC>(s, tid, 0, r##0); \
  slab_st<KC>(s, tid, 1, r##1); \
  slab_st<KC>(s, tid, 2, r##2); \
  slab_st<KC>(s, tid, 3, r##3)

// MODE: 0 plain, 1 lower tiles only (C square), 2 A lower-triangular (K range of row-block
// bi is [0,(bi+1)*TILE)), 3 A^T with A lower-triangular (K range [bi*TILE, K)), 4 = 2 under
// a name of its own for the once-per-sweep nu = L*Z launch (so that a profile lists it apart),
// tiles in an XCD-aware order; 5 = 2 under a third name: the launch that computes the rows of nu
// left over when the first rows have been computed beside the factorisation.
template <bool AT, bool BT, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g, int nbm, int nbn) {
  constexpr bool AK = AT, BKC = !BT;   // is the K index the contiguous one?
  __shared__ __attribute__((aligned(16))) double sA[2][SlabImage<AK>::SIZE];
  __shared__ __attribute__((aligned(16))) double sB[2][SlabImage<BKC>::SIZE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x, bi, bj;
  if (MODE == 1) {
    bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while ((int64_t)bi * (bi + 1) / 2 > t) --bi;
    bj = t - (int)((int64_t)bi * (bi + 1) / 2);
  } else if (MODE == 4) {
    // XCD-aware order.  Workgroups are dealt round-robin over the 8 XCDs, so t % 8 names the
    // XCD's share: each share walks 8x4 super-tiles, whose 32 tiles run together on that XCD
    // and stream the same 8 row panels of A and 4 column panels of B through its L2.
    // Longest K (largest bi) first.
    const int x = t & 7, q = t >> 3;
    const int sup = (q / (NU_ST_R * NU_ST_C)) * 8 + x, inner = q % (NU_ST_R * NU_ST_C);
    const int nsj = (nbn + NU_ST_C - 1) / NU_ST_C;
    const int sbi = sup / nsj, sbj = sup % nsj;
    bi = nbm - 1 - (sbi * NU_ST_R + inner / NU_ST_C);
    bj = sbj * NU_ST_C + inner % NU_ST_C;
    if (bi < 0 || bj >= nbn) return;
  } else if (MODE == 2 || MODE == 5) {   // longest K first
    bi = nbm - 1 - t / nbn; bj = t % nbn;
  } else {
    bi = t / nbn; bj = t % nbn;
  }
  const int z = blockIdx.y;
  const double *__restrict__ A = g.A + g.strideA * z;
  const double *__restrict__ B = g.B + g.strideB * z;
  double *C = g.C + g.strideC * z;   // may alias A or B (in-place panel and solve steps)
  const int i0 = bi * TILE, j0 = bj * TILE;
  int kbeg = 0, kend = g.K;
  if (MODE == 2 || MODE == 4 || MODE == 5) kend = min(g.K, (bi + 1) * TILE + g.k_base - z * g.k_shift);   // may be <= 0: a zero tile
  if (MODE == 3) kbeg = bi * TILE;
  const int64_t lda = g.lda, ldb = g.ldb;

  const int wm0 = (wave & 1) * 64, wn0 = (wave >> 1) * 64;
  const int lr = lane & 15, lk = lane >> 4;
  const double alpha = g.alpha, beta = g.beta;
  const int64_t ldc = g.ldc;
  // C enters through the accumulators (acc = beta/alpha * C, result = alpha * acc): its read
  // overlaps the first operand loads instead of trailing the last MFMA
  v4d acc[4][4];
  if (beta != 0.0) {
    const double sc = beta / alpha;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[tj][ti][r] = sc * C[(i0 + wm0 + ti * 16 + lr) + ldc * (int64_t)(j0 + wn0 + tj * 16 + lk + 4 * r)];
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  }

  if (kend > kbeg) {   // uniform over the workgroup
  double2 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  TILE_LOAD(AT, A, lda, i0, kbeg, ra);
  TILE_LOAD(!BT, B, ldb, j0, kbeg, rb);
  TILE_STORE(AT, sA[0], ra);
  TILE_STORE(!BT, sB[0], rb);
  __syncthreads();

  const int klast = kend - BK;
  int buf = 0;
  // fragments of k-step kk+1 are read while the 16 MFMAs of k-step kk issue
  double fa[2][4], fb[2][4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    fa[0][q] = slab_at<BKC>(sB[0], lk, wn0 + q * 16 + lr);  // MFMA row index  = j
    fb[0][q] = slab_at<AK>(sA[0], lk, wm0 + q * 16 + lr);   // MFMA col index  = i
  }
  // single-block loop body: the prefetch of the slab after the last one re-reads the last
  // slab (clamped index) and its LDS store lands in the idle buffer
  for (int k0 = kbeg; k0 < kend; k0 += BK) {
    const int kn = min(k0 + BK, klast);
    TILE_LOAD(AT, A, lda, i0, kn, ra);
    TILE_LOAD(!BT, B, ldb, j0, kn, rb);
#pragma unroll
    for (int kk = 0; kk < BK / 4; ++kk) {
      if (kk + 1 < BK / 4) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          fa[(kk + 1) & 1][q] = slab_at<BKC>(sB[buf], (kk + 1) * 4 + lk, wn0 + q * 16 + lr);
          fb[(kk + 1) & 1][q] = slab_at<AK>(sA[buf], (kk + 1) * 4 + lk, wm0 + q * 16 + lr);
        }
      } else {
        // the slab's last k-step has its fragments in registers already: stage the next
        // slab, pass the barrier and read the next slab's first fragments BEFORE issuing
        // the last 16 MFMAs, so that the barrier and the LDS latency hide under them
        TILE_STORE(AT, sA[buf ^ 1], ra);
        TILE_STORE(!BT, sB[buf ^ 1], rb);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          fa[0][q] = slab_at<BKC>(sB[buf ^ 1], lk, wn0 + q * 16 + lr);
          fb[0][q] = slab_at<AK>(sA[buf ^ 1], lk, wm0 + q * 16 + lr);
        }
      }
      __builtin_amdgcn_sched_barrier(0);   // reads first: their latency hides under the MFMAs
#pragma unroll
      for (int tj = 0; tj < 4; ++tj)
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
          acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[kk & 1][tj], fb[kk & 1][ti],
                                                             acc[tj][ti], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    buf ^= 1;
  }
  }

#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int j = j0 + wn0 + tj * 16 + lk + 4 * r;
        int i = i0 + wm0 + ti * 16 + lr;
        C[i + ldc * (int64_t)j] = alpha * acc[tj][ti][r];
      }
}

// ---------------------------------------------------------------------------------------
// Low-latency variant for products with few tiles (the panel solves and in-panel updates on
// the Cholesky's critical path, K = 128): a 128x128 tile per workgroup leaves most CUs idle
// and makes each workgroup grind through the whole depth alone.  Here a workgroup owns a
// 64x64 tile (four waves x 2x2 MFMA tiles), K advances in chunks of 64 held whole in LDS, and
// the next chunk's global loads are in flight during the current chunk's matrix work.
// ---------------------------------------------------------------------------------------
constexpr int SK = 64;       // chunk depth
constexpr int SLD = 80;      // 64 + 16: same bank argument as LDS_LD

template <bool KCONTIG>
__device__ __forceinline__ double2 chunk_ld(const double *__restrict__ base, int64_t ld, int row0,
                                            int k0, int tid, int p) {
  if (!KCONTIG)
    return *reinterpret_cast<const double2 *>(base + (row0 + ((tid & 31) << 1)) +
                                              ld * (int64_t)(k0 + (tid >> 5) + 8 * p));
  return *reinterpret_cast<const double2 *>(base + (k0 + ((tid & 31) << 1)) +
                                            ld * (int64_t)(row0 + (tid >> 5) + 8 * p));
}

constexpr int SKLD = 68;     // row stride of the index-major image of a K-contiguous operand
template <bool KCONTIG> struct ChunkImage { static constexpr int SIZE = KCONTIG ? 64 * SKLD : SK * SLD; };

template <bool KCONTIG>
__device__ __forceinline__ void chunk_st(double *s, int tid, int p, double2 v) {
  if (!KCONTIG)
    *reinterpret_cast<double2 *>(&s[((tid >> 5) + 8 * p) * SLD + ((tid & 31) << 1)]) = v;
  else
    *reinterpret_cast<double2 *>(&s[((tid >> 5) + 8 * p) * SKLD + ((tid & 31) << 1)]) = v;
}

template <bool KCONTIG>
__device__ __forceinline__ double chunk_at(const double *s, int k, int idx) {
  return KCONTIG ? s[idx * SKLD + k] : s[k * SLD + idx];
}

template <bool AT, bool BT>
__global__ __launch_bounds__(256, 2) void gemm_small_kernel(GemmArgs g, int nbn) {
  constexpr bool AK = AT, BKC = !BT;
  __shared__ __attribute__((aligned(16))) double sA[ChunkImage<AK>::SIZE];
  __shared__ __attribute__((aligned(16))) double sB[ChunkImage<BKC>::SIZE];
  // these launches sit on the factorisation's critical path and may share a SIMD with a wave of
  // a throughput launch (the chunks of nu = L*Z on the second stream): issue first
  __builtin_amdgcn_s_setprio(3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bi = blockIdx.x / nbn, bj = blockIdx.x % nbn, z = blockIdx.y;
  if (g.lower_only && bi < bj) return;   // C square: 64 x 64 tiles wholly above the diagonal are not wanted
  const double *__restrict__ A = g.A + g.strideA * z;
  const double *__restrict__ B = g.B + g.strideB * z;
  double *C = g.C + g.strideC * z;   // may alias A or B
  const int i0 = bi * 64, j0 = bj * 64;
  const int wm0 = (wave & 1) * 32, wn0 = (wave >> 1) * 32;
  const int lr = lane & 15, lk = lane >> 4;
  const double alpha = g.alpha, beta = g.beta;
  const int64_t lda = g.lda, ldb = g.ldb, ldc = g.ldc;

  double2 ra[8], rb[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    ra[p] = chunk_ld<AT>(A, lda, i0, 0, tid, p);
    rb[p] = chunk_ld<!BT>(B, ldb, j0, 0, tid, p);
  }
  v4d acc[2][2];
  if (beta != 0.0) {
    const double sc = beta / alpha;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[tj][ti][r] = sc * C[(i0 + wm0 + ti * 16 + lr) + ldc * (int64_t)(j0 + wn0 + tj * 16 + lk + 4 * r)];
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  }

  for (int k0 = 0; k0 < g.K; k0 += SK) {
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      chunk_st<AT>(sA, tid, p, ra[p]);
      chunk_st<!BT>(sB, tid, p, rb[p]);
    }
    __syncthreads();
    if (k0 + SK < g.K) {
#pragma unroll
      for (int p = 0; p < 8; ++p) {
        ra[p] = chunk_ld<AT>(A, lda, i0, k0 + SK, tid, p);
        rb[p] = chunk_ld<!BT>(B, ldb, j0, k0 + SK, tid, p);
      }
    }
#pragma unroll
    for (int kk = 0; kk < SK / 4; ++kk) {
      const double fa0 = chunk_at<BKC>(sB, kk * 4 + lk, wn0 + lr), fa1 = chunk_at<BKC>(sB, kk * 4 + lk, wn0 + 16 + lr);
      const double fb0 = chunk_at<AK>(sA, kk * 4 + lk, wm0 + lr), fb1 = chunk_at<AK>(sA, kk * 4 + lk, wm0 + 16 + lr);
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0, fb0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa0, fb1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1, fb0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa1, fb1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
  }
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(i0 + wm0 + ti * 16 + lr) + ldc * (int64_t)(j0 + wn0 + tj * 16 + lk + 4 * r)] = alpha * acc[tj][ti][r];
}

// ---------------------------------------------------------------------------------------
// Register-fed variant for A(i,k) at A[i + lda*k] (the nu = L*Z products and the trailing updates
// of the factorisation: all but a few per cent of the dense work).  No LDS image and no barrier in
// the loop: every wave takes the fragments of its own 64x64 quadrant straight from global memory
// (L1 / L2), in blocks of 8 k-values -- lane group lk holds k0 + 2 lk + {0, 1}, so that an operand
// whose K index is contiguous comes in 16-byte loads and one whose tile index is contiguous in
// 128-byte runs per 16 lanes -- and keeps the block being multiplied plus the next two in flight
// (32 registers a block; the third block is worth 2 % in the sweep, nothing in a warm loop).  The LDS version shares a slab among its four waves and pays a barrier per
// 16 k-values, which two workgroups per CU do not fully cover (0.76 of the matrix peak on the
// whole nu product); here nothing a wave waits for is owned by another wave (0.9 and more), at
// twice the L2-to-CU traffic, which that path has room for.  All address arithmetic in the loop
// is scalar: per-lane offsets are 32-bit and fixed, the block's base moves in scalar registers.
// Tile order, modes, alpha / beta and the aliasing contract are those of gemm_f64_kernel.
// ---------------------------------------------------------------------------------------
constexpr int DKB = 8;       // k-values per block
#ifndef GPIRT_DNBUF
#define GPIRT_DNBUF 3
#endif
constexpr int DNBUF_DEFAULT = GPIRT_DNBUF;   // blocks of fragments held in registers
#ifndef GPIRT_LONE_NBUF
#define GPIRT_LONE_NBUF 3
#endif
constexpr int LONE_NBUF = GPIRT_LONE_NBUF;

// Operand loads of the register-fed kernels go through raw buffer resources: a wave-uniform base
// (the wave's tile origin, four scalar registers), a fixed 32-bit offset per lane and a scalar
// offset that advances by a block of k-values.  With plain global loads the compiler rebuilt a
// 64-bit address per lane and block (a 64-bit scalar multiply for lda * k0, vector adds for the lane
// offsets: ~30 scalar and 4 vector instructions per 32 MFMAs), and that instruction stream, not the
// memory system, held the loop at 0.79 of the matrix rate; this form needs 6 scalar instructions a
// block and reaches 0.92 in the stand-alone loop (tools/sclk_probe.hip: 62.0 -> 72.6 TFLOP/s).
typedef __amdgpu_buffer_rsrc_t rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, -1, 0x00020000);   // raw, no bounds below 4 GB
}
__device__ __forceinline__ double2 buf_ld_f64x2(rsrc_t r, uint32_t lane_off, uint32_t wave_off) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  const v4u v = __builtin_amdgcn_raw_buffer_load_b128(r, lane_off, wave_off, 0);
  double2 d;
  __builtin_memcpy(&d, &v, 16);
  return d;
}

__device__ __forceinline__ double buf_ld_f64(rsrc_t r, uint32_t lane_off, uint32_t wave_off) {
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  const v2u v = __builtin_amdgcn_raw_buffer_load_b64(r, lane_off, wave_off, 0);
  double d;
  __builtin_memcpy(&d, &v, 8);
  return d;
}

// Fragment layout.  An operand whose tile index is contiguous in memory (A always; B when BT) is
// taken in 16-byte loads of TWO neighbouring rows per lane: lane lr of a 32-row group holds rows
// 2 lr and 2 lr + 1, which become the lane's entries of two MFMA tiles (the even and the odd rows of
// the group).  Half the load instructions of one row per lane per tile -- the vector memory pipe
// handles an 8-byte access at little more than half the byte rate of a 16-byte one, and with one
// row per lane it, not the matrix unit, set the pace -- and the accumulators of such a tile pair
// leave as 16-byte stores.  The sum over k of every element of C is formed in the same order.
template <bool BT, int MODE, int DNBUF>
__device__ __forceinline__ void gemm_direct_tile(const GemmArgs &g, int nbm, int nbn, int t) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  int bi, bj;
  if (MODE == 1) {
    bi = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)(bi + 1) * (bi + 2) / 2 <= t) ++bi;
    while ((int64_t)bi * (bi + 1) / 2 > t) --bi;
    bj = t - (int)((int64_t)bi * (bi + 1) / 2);
  } else if (MODE == 4) {   // XCD-aware order, see gemm_f64_kernel
    const int x = t & 7, q = t >> 3;
    const int sup = (q / (NU_ST_R * NU_ST_C)) * 8 + x, inner = q % (NU_ST_R * NU_ST_C);
    const int nsj = (nbn + NU_ST_C - 1) / NU_ST_C;
    const int sbi = sup / nsj, sbj = sup % nsj;
    bi = nbm - 1 - (sbi * NU_ST_R + inner / NU_ST_C);
    bj = sbj * NU_ST_C + inner % NU_ST_C;
    if (bi < 0 || bj >= nbn) return;
  } else if (MODE == 2 || MODE == 5) {   // longest K first
    bi = nbm - 1 - t / nbn; bj = t % nbn;
  } else {
    bi = t / nbn; bj = t % nbn;
  }
  const int z = blockIdx.y;
  const int lr = lane & 15, lk = lane >> 4;
  const int i0 = bi * TILE + (wave & 1) * 64, j0 = bj * TILE + (wave >> 1) * 64;   // this wave's quadrant
  const int64_t lda = g.lda, ldb = g.ldb, ldc = g.ldc;
  int kend = g.K;
  if (MODE == 2 || MODE == 4 || MODE == 5) kend = min(g.K, (bi + 1) * TILE + g.k_base - z * g.k_shift);   // may be <= 0: a zero tile
  const double alpha = g.alpha, beta = g.beta;
  double *C = g.C + g.strideC * z;   // may alias A or B (in-place panel and solve steps)
  // accumulator acc[tj][ti][r] of lane (lr, lk) is C(row(ti), col(tj, r)) with
  //   row(ti)    = i0 + 32 (ti / 2) + 2 lr + (ti & 1)
  //   col(tj, r) = j0 + 16 tj + lk + 4 r                              (B with K contiguous)
  //              = j0 + 32 (tj / 2) + 2 (lk + 4 r) + (tj & 1)         (BT: B's columns in pairs too)
  double *Cw = C + i0 + 2 * lr;
  auto col = [&](int tj, int r) -> int {
    return BT ? j0 + 32 * (tj >> 1) + 2 * (lk + 4 * r) + (tj & 1) : j0 + 16 * tj + lk + 4 * r;
  };

  v4d acc[4][4];
  if (beta != 0.0) {
    const double sc = beta / alpha;
#pragma unroll
    for (int tj = 0; tj < 4; ++tj)
#pragma unroll
      for (int gi = 0; gi < 2; ++gi)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const double2 v = *reinterpret_cast<const double2 *>(Cw + 32 * gi + ldc * (int64_t)col(tj, r));
          acc[tj][2 * gi][r] = sc * v.x;
          acc[tj][2 * gi + 1][r] = sc * v.y;
        }
  } else {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  }

  if (kend > 0) {   // uniform over the workgroup
    // wave-uniform bases (scalar registers) and fixed per-lane byte offsets
    const double *Aw = g.A + g.strideA * z + i0;
    const double *Bw = g.B + g.strideB * z + (BT ? (int64_t)j0 : ldb * (int64_t)j0);
    uint32_t oa[2], ob[BT ? 2 : 4];
#pragma unroll
    for (int s = 0; s < 2; ++s) oa[s] = (uint32_t)((2 * lr + lda * (2 * lk + s)) * 8);
    if (BT) {
#pragma unroll
      for (int s = 0; s < 2; ++s) ob[s] = (uint32_t)((2 * lr + ldb * (2 * lk + s)) * 8);
    } else {
#pragma unroll
      for (int tj = 0; tj < 4; ++tj) ob[tj] = (uint32_t)((2 * lk + ldb * (16 * tj + lr)) * 8);
    }
    double fa[DNBUF][2][4];
    double fb[DNBUF][2][4];      // [buffer][step][column tile]
    const int nblk = kend / DKB;
    const rsrc_t ra = make_rsrc(Aw), rb = make_rsrc(Bw);
    const uint32_t stepa = (uint32_t)(lda * (8 * DKB)), stepb = BT ? (uint32_t)(ldb * (8 * DKB)) : 8u * DKB;
    uint32_t sa = 0, sb = 0;     // byte offsets of the next block to load (scalar registers)
    int next = 0;                // its number: the blocks are loaded in order, and past the last one
                                 // the last one is loaded again (its registers are not used)
    auto load = [&](int buf, int) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int gi = 0; gi < 2; ++gi) {
          const double2 v = buf_ld_f64x2(ra, oa[s] + 256 * gi, sa);
          fa[buf][s][2 * gi] = v.x;
          fa[buf][s][2 * gi + 1] = v.y;
        }
      if (BT) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int gj = 0; gj < 2; ++gj) {
            const double2 v = buf_ld_f64x2(rb, ob[s] + 256 * gj, sb);
            fb[buf][s][2 * gj] = v.x;
            fb[buf][s][2 * gj + 1] = v.y;
          }
      } else {
#pragma unroll
        for (int tj = 0; tj < 4; ++tj) {
          const double2 v = buf_ld_f64x2(rb, ob[tj], sb);
          fb[buf][0][tj] = v.x;
          fb[buf][1][tj] = v.y;
        }
      }
      ++next;
      const bool more = next < nblk;
      sa += more ? stepa : 0u;
      sb += more ? stepb : 0u;
    };
#pragma unroll
    for (int p = 0; p < DNBUF - 1; ++p) load(p, p);
    auto mfma_block = [&](int u) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tj = 0; tj < 4; ++tj)
#pragma unroll
          for (int ti = 0; ti < 4; ++ti)
            acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[u][s][tj], fa[u][s][ti], acc[tj][ti], 0, 0, 0);
    };
    const int nmain = nblk / DNBUF * DNBUF;
    for (int blk = 0; blk < nmain; blk += DNBUF) {
#pragma unroll
      for (int u = 0; u < DNBUF; ++u) {
        load((u + DNBUF - 1) % DNBUF, blk + u + DNBUF - 1);
        __builtin_amdgcn_sched_barrier(0);   // loads first: the next blocks stay in flight under the MFMAs
        mfma_block(u);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // the blocks left over (fewer than DNBUF) are in the buffers already, in order from buffer 0
#pragma unroll
    for (int u = 0; u < DNBUF - 1; ++u)
      if (nmain + u < nblk) mfma_block(u);
  }
  // C may be an operand of this very launch: no wave writes before every wave of the tile has read
  __syncthreads();
#pragma unroll
  for (int tj = 0; tj < 4; ++tj)
#pragma unroll
    for (int gi = 0; gi < 2; ++gi)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        *reinterpret_cast<double2 *>(Cw + 32 * gi + ldc * (int64_t)col(tj, r)) =
            make_double2(alpha * acc[tj][2 * gi][r], alpha * acc[tj][2 * gi + 1][r]);
}

template <bool BT, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_direct_kernel(GemmArgs g, int nbm, int nbn) {
  gemm_direct_tile<BT, MODE, DNBUF_DEFAULT>(g, nbm, nbn, blockIdx.x);
}

// The same tiles for launches of FEWER workgroups than the chip has CUs that are to leave room for
// another stream's kernels (the look-ahead schedule of the factorisation, the rows of nu = L*Z beside
// it): a fixed grid walks the tile list (tile t, t + gridDim.x, ...).  Compiled without the
// two-waves-per-SIMD bound a workgroup holds more than half of a SIMD's registers (accumulators in
// AGPRs: ~360 of 512), so no two of them share a CU -- a grid of the standard kernel that is smaller
// than the chip is packed two workgroups to a CU and leaves half the CUs idle (224 workgroups ran at
// 29 TFLOP/s that way, 42-44 one to a CU) -- while the 64 x 64-tile kernel of the factorisation's
// in-panel updates (132 registers) still fits beside one.  Measured: 240 or 248 workgroups run the
// product faster (48 TFLOP/s) and the factorisation slower (its one-workgroup diagonal kernel waits
// for a free CU: the dispatcher binds it to a shader engine before it looks for room); a fixed grid
// of 448 workgroups of the standard kernel (two to a CU on 224 CUs) 47 TFLOP/s and the steps beside
// it slower still; four or five blocks of fragments in flight instead of three change nothing.
template <bool BT, int MODE>
__global__ __launch_bounds__(256) void gemm_direct_lone_kernel(GemmArgs g, int nbm, int nbn, int ntiles) {
#pragma nounroll
  for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
    gemm_direct_tile<BT, MODE, LONE_NBUF>(g, nbm, nbn, t);
    __syncthreads();   // the next tile's loads stay behind this tile's stores
  }
}

// The register-fed loop on 64x64 tiles for the products with few tiles and a short K (the in-panel
// updates of the factorisation, K = 128; its last, small trailing updates, K = 512): a wave owns a
// 32x32 quadrant (2x2 MFMA tiles), nothing is staged, nothing is waited for but the wave's own loads,
// and with ~100 registers a thread every workgroup of such a launch is resident at once.
template <bool BT>
__global__ __launch_bounds__(256) void gemm_direct_small_kernel(GemmArgs g, int nbn) {
  __builtin_amdgcn_s_setprio(3);   // critical path of the factorisation, see gemm_small_kernel
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int bi = blockIdx.x / nbn, bj = blockIdx.x % nbn, z = blockIdx.y;
  if (g.lower_only && bi < bj) return;   // C square: 64 x 64 tiles wholly above the diagonal are not wanted
  const int lr = lane & 15, lk = lane >> 4;
  const int i0 = bi * 64 + (wave & 1) * 32, j0 = bj * 64 + (wave >> 1) * 32;
  const int64_t lda = g.lda, ldb = g.ldb, ldc = g.ldc;
  const double alpha = g.alpha, beta = g.beta;
  double *C = g.C + g.strideC * z;
  v4d acc[2][2];
  if (beta != 0.0) {
    const double sc = beta / alpha;
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[tj][ti][r] = sc * C[(i0 + ti * 16 + lr) + ldc * (int64_t)(j0 + tj * 16 + lk + 4 * r)];
  } else {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) acc[a][b] = (v4d){0.0, 0.0, 0.0, 0.0};
  }
  const double *Aw = g.A + g.strideA * z + i0;
  const double *Bw = g.B + g.strideB * z + (BT ? (int64_t)j0 : ldb * (int64_t)j0);
  uint32_t oa[2], ob[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) oa[s] = (uint32_t)((lr + lda * (2 * lk + s)) * 8);
  if (BT) {
#pragma unroll
    for (int s = 0; s < 2; ++s) ob[s] = (uint32_t)((lr + ldb * (2 * lk + s)) * 8);
  } else {
#pragma unroll
    for (int tj = 0; tj < 2; ++tj) ob[tj] = (uint32_t)((2 * lk + ldb * (16 * tj + lr)) * 8);
  }
  constexpr int NB = 4;
  double fa[NB][2][2], fb[NB][2][2];
  const int nblk = g.K / DKB;
  // buffer resources and a scalar offset that advances by a block, as in gemm_direct_tile
  const rsrc_t ra = make_rsrc(Aw), rb = make_rsrc(Bw);
  const uint32_t stepa = (uint32_t)(lda * (8 * DKB)), stepb = BT ? (uint32_t)(ldb * (8 * DKB)) : 8u * DKB;
  uint32_t sa = 0, sb = 0;
  int next = 0;
  auto load = [&](int buf, int) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ti = 0; ti < 2; ++ti) fa[buf][s][ti] = buf_ld_f64(ra, oa[s] + 128 * ti, sa);
    if (BT) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) fb[buf][s][tj] = buf_ld_f64(rb, ob[s] + 128 * tj, sb);
    } else {
#pragma unroll
      for (int tj = 0; tj < 2; ++tj) {
        const double2 v = buf_ld_f64x2(rb, ob[tj], sb);
        fb[buf][0][tj] = v.x;
        fb[buf][1][tj] = v.y;
      }
    }
    ++next;
    const bool more = next < nblk;
    sa += more ? stepa : 0u;
    sb += more ? stepb : 0u;
  };
  auto mfma_block = [&](int u) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
          acc[tj][ti] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[u][s][tj], fa[u][s][ti], acc[tj][ti], 0, 0, 0);
  };
#pragma unroll
  for (int p = 0; p < NB - 1; ++p) load(p, p);
  const int nmain = nblk / NB * NB;
  for (int blk = 0; blk < nmain; blk += NB) {
#pragma unroll
    for (int u = 0; u < NB; ++u) {
      load((u + NB - 1) % NB, blk + u + NB - 1);
      __builtin_amdgcn_sched_barrier(0);
      mfma_block(u);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#pragma unroll
  for (int u = 0; u < NB - 1; ++u)
    if (nmain + u < nblk) mfma_block(u);
  __syncthreads();   // C may be an operand of this launch
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        C[(i0 + ti * 16 + lr) + ldc * (int64_t)(j0 + tj * 16 + lk + 4 * r)] = alpha * acc[tj][ti][r];
}

template <bool BT>
static int launch_direct_mode(hipStream_t st, const GemmArgs &a, int mode, dim3 grid, int nbm, int nbn) {
  dim3 block(256);
  switch (mode) {
    case 0: hipLaunchKernelGGL((gemm_direct_kernel<BT, 0>), grid, block, 0, st, a, nbm, nbn); break;
    case 1: hipLaunchKernelGGL((gemm_direct_kernel<BT, 1>), grid, block, 0, st, a, nbm, nbn); break;
    case 2: hipLaunchKernelGGL((gemm_direct_kernel<BT, 2>), grid, block, 0, st, a, nbm, nbn); break;
    case 4: hipLaunchKernelGGL((gemm_direct_kernel<BT, 4>), grid, block, 0, st, a, nbm, nbn); break;
    default: hipLaunchKernelGGL((gemm_direct_kernel<BT, 5>), grid, block, 0, st, a, nbm, nbn); break;
  }
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// GPIRT_GEMM_LDS=1 keeps every product on the LDS-staged kernel (for comparisons)
static bool gemm_lds_only() {
  static const bool v = getenv("GPIRT_GEMM_LDS") && atoi(getenv("GPIRT_GEMM_LDS")) != 0;
  return v;
}

template <bool AT, bool BT>
static int launch_gemm_mode(hipStream_t st, const GemmArgs &a, int mode, dim3 grid, int nbm, int nbn) {
  dim3 block(256);
  switch (mode) {
    case 0: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, 0>), grid, block, 0, st, a, nbm, nbn); break;
    case 1: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, 1>), grid, block, 0, st, a, nbm, nbn); break;
    case 2: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, 2>), grid, block, 0, st, a, nbm, nbn); break;
    case 4: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, 4>), grid, block, 0, st, a, nbm, nbn); break;
    case 5: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, 5>), grid, block, 0, st, a, nbm, nbn); break;
    default: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, 3>), grid, block, 0, st, a, nbm, nbn); break;
  }
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// lower-tiles-only launches with at most this many 128 x 128 tiles go to the 64 x 64 kernel
static int small_lower_tiles() {
  static const int v = getenv("GPIRT_SMALL_LOWER") ? atoi(getenv("GPIRT_SMALL_LOWER")) : 300;
  return v;
}

// the register-fed kernels take rows in 16-byte pairs: every operand 16-byte aligned, even strides;
// and their operand loads address a tile's K range with 32-bit byte offsets off the tile's origin
static bool pairs_aligned(const GemmArgs &a) {
  const uintptr_t p = (uintptr_t)a.A | (uintptr_t)a.B | (uintptr_t)a.C;
  const int64_t l = a.lda | a.ldb | a.ldc | a.strideA | a.strideB | a.strideC;
  const int64_t reach = 8 * (std::max(a.lda, a.ldb) * ((int64_t)a.K + 144) + 256);
  return p % 16 == 0 && l % 2 == 0 && reach < ((int64_t)1 << 32);
}

// workgroups of a fixed-grid launch: one per CU on 7 of every 8 CUs, so that every shader engine
// keeps a CU free (the dispatcher binds a waiting workgroup to a shader engine before it looks for room)
static int persist_workgroups() {
  static const int v = getenv("GPIRT_PERSIST_WGS") ? std::max(1, atoi(getenv("GPIRT_PERSIST_WGS"))) : 224;
  return v;
}

int launch_gemm(hipStream_t st, bool AT, bool BT, const GemmArgs &a) {
  if (a.M <= 0 || a.N <= 0 || a.batch <= 0) return GPIRT_OK;
  if (a.M % TILE || a.N % TILE || a.K % BK || a.K <= 0) {
    set_error("launch_gemm: M=%d N=%d K=%d not tile multiples", a.M, a.N, a.K);
    return GPIRT_ERR_BAD_ARG;
  }
  int nbm = a.M / TILE, nbn = a.N / TILE;
  int mode = a.lower_only ? 1 : (a.k_trim == 1 ? 2 : (a.k_trim == 2 ? 3 : (a.k_trim == 3 ? 4 : (a.k_trim == 4 ? 5 : 0))));
  int64_t tiles = a.lower_only ? (int64_t)nbm * (nbm + 1) / 2 : (int64_t)nbm * nbn;
  // lone: a fixed grid of at most persist_workgroups() workgroups, one to a CU (the small 64 x 64
  // variant below still takes the launches with few tiles)
  // (the triangular product also in slabs of K side by side, the batch index: the grid's share per slab)
  const bool lone = a.lone && (a.batch == 1 || ((mode == 2 || mode == 5) && !BT && a.batch <= 16)) && !AT && !gemm_lds_only() &&
                    pairs_aligned(a) && a.k_base % BK == 0 && a.k_shift % BK == 0 &&
                    a.lda * 4 * 8 < (int64_t)1 << 31 && a.ldb * 80 * 8 < (int64_t)1 << 31 &&
                    ((mode == 1 && BT && tiles > small_lower_tiles()) || ((mode == 2 || mode == 5) && !BT) ||
                     (mode == 0 && !BT && tiles > SMALL_GEMM_TILES));
  if (mode == 4) {   // whole 8x4 super-tiles, a multiple of 8 of them (one share per XCD)
    int64_t nsup = (int64_t)((nbm + NU_ST_R - 1) / NU_ST_R) * ((nbn + NU_ST_C - 1) / NU_ST_C);
    tiles = (nsup + 7) / 8 * 8 * NU_ST_R * NU_ST_C;
  }
  // few tiles and a depth that is a multiple of the chunk: the low-latency 64x64 variant.  The
  // late trailing updates of the factorisation (lower tiles only, depth 512) have fewer tiles
  // than the chip has workgroup slots: a 128 x 128 tile of that depth runs ~80 us however few
  // there are, four 64 x 64 tiles side by side a third of that.
  const int small_lower = small_lower_tiles();
  if ((mode == 0 && tiles * a.batch <= SMALL_GEMM_TILES && a.K % SK == 0 && a.K <= 512) ||
      (mode == 1 && !AT && BT && tiles * a.batch <= small_lower && a.K % SK == 0 && a.K <= 512)) {
    if (mode == 1) tiles = (int64_t)nbm * nbn;      // the grid covers the square; tiles above the diagonal exit
    dim3 sgrid((unsigned)(4 * tiles), (unsigned)a.batch), block(256);
    static const bool small_direct = !(getenv("GPIRT_SMALL_DIRECT") && atoi(getenv("GPIRT_SMALL_DIRECT")) == 0);
    const bool direct_ok = small_direct && !gemm_lds_only() && a.lda * 4 * 8 < (int64_t)1 << 31 && a.ldb * 80 * 8 < (int64_t)1 << 31 &&
                           8 * (std::max(a.lda, a.ldb) * ((int64_t)a.K + 144) + 256) < ((int64_t)1 << 32);
    if (!AT && BT && direct_ok) hipLaunchKernelGGL((gemm_direct_small_kernel<true>), sgrid, block, 0, st, a, 2 * nbn);
    else if (!AT && !BT && direct_ok) hipLaunchKernelGGL((gemm_direct_small_kernel<false>), sgrid, block, 0, st, a, 2 * nbn);
    else if (!AT && BT) hipLaunchKernelGGL((gemm_small_kernel<false, true>), sgrid, block, 0, st, a, 2 * nbn);
    else if (!AT && !BT) hipLaunchKernelGGL((gemm_small_kernel<false, false>), sgrid, block, 0, st, a, 2 * nbn);
    else if (AT && !BT) hipLaunchKernelGGL((gemm_small_kernel<true, false>), sgrid, block, 0, st, a, 2 * nbn);
    else { set_error("launch_gemm: A^T B^T is not built"); return GPIRT_ERR_BAD_ARG; }
    GP_HIP(hipGetLastError());
    return GPIRT_OK;
  }
  if (lone) {
    // the rounds the tile list takes with the largest grid, then the smallest grid that needs no more:
    // the same time, and more CUs left for the kernels beside it
    const int64_t share = std::max(1, persist_workgroups() / a.batch);
    const int64_t rounds = (tiles + share - 1) / share;
    const dim3 lgrid((unsigned)((tiles + rounds - 1) / rounds), (unsigned)a.batch);
    if (mode == 1) hipLaunchKernelGGL((gemm_direct_lone_kernel<true, 1>), lgrid, dim3(256), 0, st, a, nbm, nbn, (int)tiles);
    else if (mode == 2 || mode == 5) hipLaunchKernelGGL((gemm_direct_lone_kernel<false, 2>), lgrid, dim3(256), 0, st, a, nbm, nbn, (int)tiles);
    else hipLaunchKernelGGL((gemm_direct_lone_kernel<false, 0>), lgrid, dim3(256), 0, st, a, nbm, nbn, (int)tiles);
    GP_HIP(hipGetLastError());
    return GPIRT_OK;
  }
  dim3 grid((unsigned)tiles, (unsigned)a.batch);
  static_assert(BK % DKB == 0, "K granularity");
  if (!AT && !gemm_lds_only() && a.k_base % BK == 0 && a.k_shift % BK == 0 && a.lda * 4 * 8 < (int64_t)1 << 31 && a.ldb * 80 * 8 < (int64_t)1 << 31 && pairs_aligned(a)) {
    if (!BT && (mode == 0 || mode == 2 || mode == 4 || mode == 5)) return launch_direct_mode<false>(st, a, mode, grid, nbm, nbn);
    if (BT && (mode == 0 || mode == 1)) return launch_direct_mode<true>(st, a, mode, grid, nbm, nbn);
  }
  // only the combinations the sweep uses are instantiated
  if (!AT && !BT && (mode == 0 || mode == 2 || mode == 4 || mode == 5)) return launch_gemm_mode<false, false>(st, a, mode, grid, nbm, nbn);
  if (!AT && BT && (mode == 0 || mode == 1)) return launch_gemm_mode<false, true>(st, a, mode, grid, nbm, nbn);
  if (AT && !BT && (mode == 0 || mode == 3)) return launch_gemm_mode<true, false>(st, a, mode, grid, nbm, nbn);
  set_error("launch_gemm: combination AT=%d BT=%d mode=%d is not built", (int)AT, (int)BT, mode);
  return GPIRT_ERR_BAD_ARG;
}

// ---------------------------------------------------------------------------------------
// Kernel-matrix build (a1) and the grid cross-covariance K(x, xs) used by f* (a6)
// ---------------------------------------------------------------------------------------
constexpr int KB_COLS = 8;   // columns per thread: the row's input is loaded once, the grid is 8x smaller

__global__ void k_build_kernel(const double *__restrict__ x, int64_t x_stride, int n, int n_pad,
                               double inv_ls2, double sf2, double jitter, double *__restrict__ S,
                               int full) {
  const int z = blockIdx.z;
  const double *xp = x + x_stride * z;
  double *Sp = S + (int64_t)n_pad * n_pad * z;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int j0 = blockIdx.y * KB_COLS;          // KB_COLS divides TILE: the columns share a tile column
  if (i >= n_pad) return;
  // the lower form leaves the tiles wholly above the diagonal alone: nothing reads them (the
  // factorisation and the products with the factor touch tiles on and below the diagonal only)
  if (!full && i < (j0 & ~(TILE - 1))) return;
  const double xi = i < n ? xp[i] : 0.0;
#pragma unroll
  for (int q = 0; q < KB_COLS; ++q) {
    const int j = j0 + q;
    double v;
    if (i < n && j < n) {
      if (i >= j || full) {
        double d = xi - xp[j];
        v = sf2 * exp(-0.5 * d * d * inv_ls2) + (i == j ? jitter : 0.0);
      } else v = 0.0;
    } else v = (i == j) ? 1.0 : 0.0;
    Sp[i + (int64_t)n_pad * j] = v;
  }
}

int launch_k_build(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                   double ls, double sf, double jitter, double *S, bool full) {
  static_assert(TILE % KB_COLS == 0, "columns per thread");
  dim3 block(256), grid((n_pad + 255) / 256, n_pad / KB_COLS, batch);
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

// Kc[r + n_pad*j] = sf^2 exp(-(c_j - x_r)^2 / (2 ls^2)) for the nodes c_0..c_{R-1}, zero outside
__global__ void kstar_nodes_kernel(const double *__restrict__ x, int64_t x_stride, int n, int n_pad,
                                   double inv_ls2, double sf2, const double *__restrict__ nodes, int R,
                                   double *__restrict__ Kc, int64_t group_stride) {
  const int z = blockIdx.z;
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  int j = blockIdx.y;
  if (r >= n_pad) return;
  double v = 0.0;
  if (r < n && j < R) {
    double d = nodes[j] - x[x_stride * z + r];
    v = sf2 * exp(-0.5 * d * d * inv_ls2);
  }
  Kc[group_stride * z + r + (int64_t)n_pad * j] = v;
}

int launch_kstar_nodes(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                       double ls, double sf, const double *nodes, int R, int R_pad, double *Kc,
                       int64_t group_stride) {
  dim3 block(256), grid((n_pad + 255) / 256, R_pad, batch);
  hipLaunchKernelGGL(kstar_nodes_kernel, grid, block, 0, st, x, x_stride, n, n_pad, 1.0 / (ls * ls),
                     sf * sf, nodes, R, Kc, group_stride);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// Barycentric Lagrange basis of the Chebyshev nodes at every input: Pm[r + n_pad*j] = l_j(x_r)
// (rows r >= n zero).  One wave per row.  An input outside [lo, hi] would be extrapolated:
// it sets *outside instead (the caller then takes another route).
__global__ __launch_bounds__(256) void bary_kernel(const double *__restrict__ x, int64_t x_stride, int n,
                                                   int n_pad, const double *__restrict__ nodes,
                                                   const double *__restrict__ bw, int R, double lo,
                                                   double hi, double *__restrict__ Pm,
                                                   int64_t group_stride, int *__restrict__ outside) {
  const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6), z = blockIdx.y;
  if (r >= n_pad) return;
  double *row = Pm + group_stride * z + r;
  if (r >= n) {
    for (int j = lane; j < R; j += 64) row[(int64_t)n_pad * j] = 0.0;
    return;
  }
  const double xr = x[x_stride * z + r];
  if (!(xr >= lo && xr <= hi) && lane == 0) atomicOr(outside, 1);
  double den = 0.0;
  int hit = -1;
  for (int j = lane; j < R; j += 64) {
    const double d = xr - nodes[j];
    if (d == 0.0) hit = j;
    else den += bw[j] / d;
  }
  den = wave_sum(den);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) hit = max(hit, __shfl_xor(hit, off, 64));
  const double inv = 1.0 / den;
  for (int j = lane; j < R; j += 64) {
    const double d = xr - nodes[j];
    row[(int64_t)n_pad * j] = hit >= 0 ? (j == hit ? 1.0 : 0.0) : (bw[j] / d) * inv;
  }
}

int launch_bary(hipStream_t st, const double *x, int64_t x_stride, int n, int n_pad, int batch,
                const double *nodes, const double *bw, int R, double lo, double hi, double *Pm,
                int64_t group_stride, int *outside) {
  hipLaunchKernelGGL(bary_kernel, dim3((n_pad + 3) / 4, batch), dim3(256), 0, st, x, x_stride, n, n_pad,
                     nodes, bw, R, lo, hi, Pm, group_stride, outside);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// A[(row0 + i) + ld*j] = (i == j) ? v : 0 for i, j < TILE, `batch` groups `stride` apart
__global__ void scaled_identity_kernel(double *__restrict__ A, int64_t ld, int row0, int64_t stride, double v) {
  const int i = threadIdx.x, j = blockIdx.x;
  A[stride * blockIdx.y + row0 + i + ld * j] = i == j ? v : 0.0;
}

int launch_scaled_identity(hipStream_t st, double *A, int64_t ld, int row0, int batch, int64_t stride,
                           double v) {
  hipLaunchKernelGGL(scaled_identity_kernel, dim3(TILE, batch), dim3(TILE), 0, st, A, ld, row0, stride, v);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// J[b] = s2 I + sum of the Gram slabs of the blocks before b (exclusive prefix over b), TILE x TILE each
__global__ void prefix_gram_kernel(const double *__restrict__ slabs, int nblk, double s2,
                                   double *__restrict__ J) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;      // element of a TILE x TILE block
  const double d = (e % TILE == e / TILE) ? s2 : 0.0;
  double acc = 0.0;
  for (int b = 0; b < nblk; ++b) {
    J[(int64_t)TILE * TILE * b + e] = acc + d;
    acc += slabs[(int64_t)TILE * TILE * b + e];
  }
}

int launch_prefix_gram(hipStream_t st, const double *slabs, int nblk, double s2, double *J) {
  hipLaunchKernelGGL(prefix_gram_kernel, dim3(TILE * TILE / 256), dim3(256), 0, st, slabs, nblk, s2, J);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// zero above the diagonal of `batch` TILE x TILE blocks (ld = TILE)
__global__ void tril_kernel(double *__restrict__ A) {
  const int i = threadIdx.x, j = blockIdx.x;
  if (i < j) A[(int64_t)TILE * TILE * blockIdx.y + i + TILE * j] = 0.0;
}

int launch_tril(hipStream_t st, double *A, int batch) {
  hipLaunchKernelGGL(tril_kernel, dim3(TILE, batch), dim3(TILE), 0, st, A);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// in place, over nblk consecutive blocks of `elems` doubles: V[b] <- V[0] + ... + V[b-1]
__global__ void prefix_blocks_kernel(double *__restrict__ V, int64_t elems, int nblk) {
  const int64_t e = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (e >= elems) return;
  double2 acc = make_double2(0.0, 0.0);
  for (int b = 0; b < nblk; ++b) {
    double2 *p = reinterpret_cast<double2 *>(V + elems * b + e);
    const double2 v = *p;
    *p = acc;
    acc.x += v.x; acc.y += v.y;
  }
}

int launch_prefix_blocks(hipStream_t st, double *V, int64_t elems, int nblk) {
  hipLaunchKernelGGL(prefix_blocks_kernel, dim3((unsigned)((elems / 2 + 255) / 256)), dim3(256), 0, st, V,
                     elems, nblk);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// defined in potrf_diag.hip
int launch_potrf_diag(hipStream_t st, double *S, int n_pad, int kb, int batch, double *inv_diag, int nb,
                      int *d_info, bool offdiag);
int launch_trsm_panel(hipStream_t st, double *S, int n_pad, int kb, int batch, const double *inv_diag);
int launch_block_inverses(hipStream_t st, const double *S, int n_pad, int batch, double *inv_diag);
int launch_potrf_rawcopy(hipStream_t st, const double *S, int n_pad, int kb, int oe, int batch, double *raw);
int launch_potrf_inpanel(hipStream_t st, double *S, int n_pad, int kb, int o, int oe, int batch,
                         const double *inv_diag, double *raw);
int launch_potrf_diag_tall(hipStream_t st, double *S, int n_pad, int kb, int batch, double *inv_diag, int *d_info,
                           int o, int oe, int c);
int launch_potrf_tall_step(hipStream_t st, double *S, int n_pad, int o, int oe, int c, int batch, const double *inv_diag);

// Off-diagonal blocks of the inverted diagonal super-blocks by block substitution: for a
// lower-triangular [[A,0],[B,C]] the inverse is [[A^-1,0],[-C^-1 B A^-1, C^-1]].
static int build_super_inverses(hipStream_t st, const double *L, int n_pad, double *inv, double *work) {
  const int nb = n_pad / TILE, nfull = nb / 4, tail = nb % 4;
  // bottom = [top0+tsz, top0+tsz+bsz) gets -inv(bottom) * L[bottom, top] * inv(top), for `count`
  // consecutive super-blocks starting at c0 (offsets in rows inside the super-block)
  auto levelup = [&](int c0, int count, int top0, int tsz, int bsz) -> int {
    if (count <= 0) return GPIRT_OK;
    const int bot0 = top0 + tsz;
    const double *Lc = L + (int64_t)c0 * SUP * (n_pad + 1);
    double *Ic = inv + (int64_t)c0 * SUP * SUP;
    double *Wc = work + (int64_t)c0 * 256 * 256;
    GemmArgs t{};   // T = L[bottom, top] * inv(top)
    t.A = Lc + bot0 + (int64_t)n_pad * top0; t.lda = n_pad; t.strideA = (int64_t)SUP * (n_pad + 1);
    t.B = Ic + (int64_t)top0 * (SUP + 1); t.ldb = SUP; t.strideB = (int64_t)SUP * SUP;
    t.C = Wc; t.ldc = 256; t.strideC = 256 * 256;
    t.M = bsz; t.N = tsz; t.K = tsz; t.batch = count; t.alpha = 1.0; t.beta = 0.0;
    GP_TRY(launch_gemm(st, false, false, t));
    GemmArgs x{};   // inv[bottom, top] = -inv(bottom) * T
    x.A = Ic + (int64_t)bot0 * (SUP + 1); x.lda = SUP; x.strideA = (int64_t)SUP * SUP;
    x.B = Wc; x.ldb = 256; x.strideB = 256 * 256;
    x.C = Ic + bot0 + (int64_t)SUP * top0; x.ldc = SUP; x.strideC = (int64_t)SUP * SUP;
    x.M = bsz; x.N = tsz; x.K = bsz; x.batch = count; x.alpha = -1.0; x.beta = 0.0;
    return launch_gemm(st, false, false, x);
  };
  GP_TRY(levelup(0, nfull, 0, TILE, TILE));
  GP_TRY(levelup(0, nfull, 2 * TILE, TILE, TILE));
  GP_TRY(levelup(0, nfull, 0, 2 * TILE, 2 * TILE));
  if (tail >= 2) GP_TRY(levelup(nfull, 1, 0, TILE, TILE));
  if (tail == 3) GP_TRY(levelup(nfull, 1, 0, 2 * TILE, TILE));
  return GPIRT_OK;
}

int launch_super_inverses(hipStream_t st, const double *L, int n_pad, int batch, double *inv, double *work,
                          bool complete_blocks) {
  // after a factorisation with POTRF_SUBSTITUTION the image holds the inverted 32 x 32 diagonal
  // blocks only: the 128 x 128 inverses come first
  if (complete_blocks) GP_TRY(launch_block_inverses(st, L, n_pad, batch, inv));
  if (n_pad / TILE > 1)
    for (int z = 0; z < batch; ++z)
      GP_TRY(build_super_inverses(st, L + (int64_t)n_pad * n_pad * z, n_pad, inv + inv_group_elems(n_pad) * z,
                                  work + (int64_t)inv_nsup(n_pad) * 256 * 256 * z));
  return GPIRT_OK;
}

// Right-looking blocked Cholesky with two-level blocking: a super-panel of 4 block columns is
// factored by 128-wide steps (diagonal kernel, panel solve, update of the rest of the
// super-panel); the trailing matrix then gets ONE rank-512 update (lower tiles only), which cuts
// the read-modify-write traffic on it four-fold.  With a side stream (PotrfHook::side) that
// update is split: the four block columns the next super-panel's steps need stay on `st`, the
// rest runs on the side stream beside those steps as a fixed grid that leaves CUs free -- the
// one-workgroup diagonal kernel needs a whole CU, and beside an update that fills every CU it
// only starts when the update drains.  (Left-looking steps inside a super-panel -- each block
// column updated by all earlier ones in one product right before it is factored, launches a
// third as wide -- were measured and change nothing: these launches cost what their first loads
// and their stores cost, not what their tiles cost.)
int launch_potrf(hipStream_t st, double *S, int n_pad, int batch, double *inv, int *d_info,
                 double *work, const PotrfHook *hook, int mode) {
  const int nb = n_pad / TILE;
  static const bool no_subst = getenv("GPIRT_NO_SUBSTITUTION") != nullptr;
  const bool subst = (mode & POTRF_SUBSTITUTION) && !no_subst;
  const int64_t sS = (int64_t)n_pad * n_pad, sI = inv_group_elems(n_pad);
  auto panel_solve = [&](int k) -> int {   // L_ik = A_ik * inv(L_kk)^T, i > k
    if (subst) return launch_trsm_panel(st, S, n_pad, k, batch, inv);
    int rem = nb - 1 - k;
    double *panel = S + (int64_t)(k + 1) * TILE + (int64_t)n_pad * k * TILE;
    GemmArgs p{};
    p.A = panel; p.lda = n_pad; p.strideA = sS;
    p.B = inv + inv_block_offset(k); p.ldb = SUP; p.strideB = sI;
    p.C = panel; p.ldc = n_pad; p.strideC = sS;
    p.M = rem * TILE; p.N = TILE; p.K = TILE; p.batch = batch; p.alpha = 1.0; p.beta = 0.0;
    return launch_gemm(st, false, true, p);
  };
  auto update_rect = [&](int k, int c0, int c1) -> int {   // cols [c0,c1) by panel k, rows >= c0
    if (c1 <= c0) return GPIRT_OK;
    GemmArgs u{};
    u.A = S + (int64_t)c0 * TILE + (int64_t)n_pad * k * TILE; u.lda = n_pad; u.strideA = sS;
    u.B = u.A; u.ldb = n_pad; u.strideB = sS;
    u.C = S + (int64_t)c0 * TILE * (n_pad + 1); u.ldc = n_pad; u.strideC = sS;
    u.M = (nb - c0) * TILE; u.N = (c1 - c0) * TILE; u.K = TILE; u.batch = batch;
    u.alpha = -1.0; u.beta = 1.0;
    return launch_gemm(st, false, true, u);
  };
  // the rank-(oe-o)*128 update of the lower tiles from block (r0, r0) on by block columns [o, oe)
  auto trailing = [&](int o, int oe, int r0, int fixed_grid) -> GemmArgs {
    GemmArgs u{};
    u.A = S + (int64_t)r0 * TILE + (int64_t)n_pad * o * TILE; u.lda = n_pad; u.strideA = sS;
    u.B = u.A; u.ldb = n_pad; u.strideB = sS;
    u.C = S + (int64_t)r0 * TILE * (n_pad + 1); u.ldc = n_pad; u.strideC = sS;
    u.M = (nb - r0) * TILE; u.N = u.M; u.K = (oe - o) * TILE; u.batch = batch;
    u.alpha = -1.0; u.beta = 1.0; u.lower_only = 1; u.lone = fixed_grid;
    return u;
  };
  const bool ahead = hook && hook->side && hook->ev && batch == 1;
  // The steps of a super-panel with the diagonal super-block FIRST (potrf_diag.hip).  The chain of small kernels
  // costs what its launches' first loads and stores cost, whatever the number of rows, so per 128 columns it
  // only touches the rows of the diagonal super-block: the diagonal kernel, then ONE launch for panel solve +
  // update inside the super-block (potrf_inpanel_kernel).  The rows below the super-block take block column k
  // (solve, update of the super-panel's later block columns) as extra workgroups of the NEXT diagonal block's
  // launch: that step needs nothing of the chain beyond diagonal block k and the step inside the super-block,
  // and the chain does not wait for it; only the last block column's step, the solve alone, is a launch of its
  // own before the trailing update.  Measured at n = 10 000 (factorisation alone / with look-ahead / sweeps per
  // second in the default schedule): step-by-step over all rows, three launches per step, 10.2 ms / 8.8 ms /
  // 55.2; this form 9.6 / 8.4 / 56.3.  Also measured and not kept: all rows below in ONE launch per super-panel
  // (a row block's chain over the four block columns in registers: 87 us per launch whatever the rows, 2112
  // dependent MFMAs per wave: 10.9 ms alone), and the steps of the rows below as launches of their own on a
  // third stream (a 152-register kernel that fits beside the fixed-grid launches of the second stream: the
  // steps inside the super-block then share the free CUs with it and slow from 20 to 32 us: 56.2).
  // GPIRT_POTRF_STEPWISE=1 selects the step-by-step form.
  static const bool stepwise = getenv("GPIRT_POTRF_STEPWISE") && atoi(getenv("GPIRT_POTRF_STEPWISE")) != 0;
  const bool block_first = subst && !stepwise && work != nullptr;
  for (int o = 0; o < nb; o += 4) {
    const int oe = min(o + 4, nb), sp = o / 4;
    if (block_first) {
      GP_TRY(launch_potrf_rawcopy(st, S, n_pad, o, oe, batch, work));
      for (int k = o; k < oe; ++k) {
        GP_TRY(launch_potrf_diag_tall(st, S, n_pad, k, batch, inv, d_info, o, oe, k > o ? k - 1 - o : -1));
        GP_TRY(launch_potrf_inpanel(st, S, n_pad, k, o, oe, batch, inv, work));
      }
      GP_TRY(launch_potrf_tall_step(st, S, n_pad, o, oe, oe - 1 - o, batch, inv));
    } else {
      for (int k = o; k < oe; ++k) {
        GP_TRY(launch_potrf_diag(st, S, n_pad, k, batch, inv, nb, d_info, !subst));
        if (k + 1 < nb) GP_TRY(panel_solve(k));
        GP_TRY(update_rect(k, k + 1, oe));
      }
    }
    if (oe < nb && !ahead) {
      GemmArgs u = trailing(o, oe, oe, 0);
      const double r = (double)(nb - oe);
      const double flop = r * (r + 1.0) * 0.5 * 2.0 * TILE * TILE * (double)u.K * batch;   // lower tiles only
      // timed: the launches of the 128 x 128-tile kernel (the last, small updates run on 64 x 64 tiles)
      const bool timed = hook && hook->timer && r * (r + 1.0) * 0.5 * batch > small_lower_tiles();
      if (timed) GP_TRY(hook->timer(hook->ctx, 1, flop, st));
      GP_TRY(launch_gemm(st, false, true, u));
      if (timed) GP_TRY(hook->timer(hook->ctx, 0, flop, st));
    }
    if (oe < nb && ahead) {
      // look-ahead: block columns [oe, oe + 4) are all that the next chain needs
      const int pe = min(oe + 4, nb);
      if (pe < nb) {
        GP_HIP(hipEventRecord(hook->ev[2 * sp], st));              // panel sp is final
        GP_HIP(hipStreamWaitEvent(hook->side, hook->ev[2 * sp], 0));
        GemmArgs u = trailing(o, oe, pe, 1);
        const double r = (double)(nb - pe);
        const double flop = r * (r + 1.0) * 0.5 * 2.0 * TILE * TILE * (double)u.K;
        const bool timed = hook->timer && r * (r + 1.0) * 0.5 > small_lower_tiles();
        if (timed) GP_TRY(hook->timer(hook->ctx, 1, flop, hook->side));
        GP_TRY(launch_gemm(hook->side, false, true, u));
        if (timed) GP_TRY(hook->timer(hook->ctx, 0, flop, hook->side));
        GP_HIP(hipEventRecord(hook->ev[2 * sp + 1], hook->side));  // the side stream's share of update sp is done
      }
      if (hook->fn) GP_TRY(hook->fn(hook->ctx, o, oe));
      // the strip the next chain needs: it was last written by the side stream's share of update sp - 1
      if (sp > 0 && min(o + 4, nb) < nb) GP_HIP(hipStreamWaitEvent(st, hook->ev[2 * (sp - 1) + 1], 0));
      GemmArgs p{};
      p.A = S + (int64_t)oe * TILE + (int64_t)n_pad * o * TILE; p.lda = n_pad; p.strideA = sS;
      p.B = p.A; p.ldb = n_pad; p.strideB = sS;
      p.C = S + (int64_t)oe * TILE * (n_pad + 1); p.ldc = n_pad; p.strideC = sS;
      p.M = (nb - oe) * TILE; p.N = (pe - oe) * TILE; p.K = (oe - o) * TILE; p.batch = 1;
      p.alpha = -1.0; p.beta = 1.0;
      // The strip and the side stream's share of the update become runnable at the same moment, and
      // workgroups of the 128 x 128 kernel do not fit beside lone ones: the strip is dispatched first
      // on the boxes measured (140 us); second, it would crawl on the CUs the fixed grid leaves free
      // (560 us, seen with another launch order).  On 64 x 64 tiles, which fit either way, it costs
      // 1-1.5 % of the sweep, so the large strips stay on 128 x 128 tiles; a box where the order
      // flips loses the schedule's gain, and the sampler's own timing then keeps the plain one.
      GP_TRY(launch_gemm(st, false, true, p));
      continue;
    }
    if (hook && hook->fn) GP_TRY(hook->fn(hook->ctx, o, oe));
  }
  if (!(mode & POTRF_DEFER_INVERSES)) GP_TRY(launch_super_inverses(st, S, n_pad, batch, inv, work, subst));
  return GPIRT_OK;
}

// out[r + ldo*j] = sum over slabs of x[slab][r + ldx*j], two rows per thread
__global__ void sum_slabs_kernel(const double *__restrict__ x, int nslab, int64_t slab_stride, int rows,
                                 int64_t ldx, double *__restrict__ out, int64_t ldo) {
  const int r = 2 * (blockIdx.x * blockDim.x + threadIdx.x), j = blockIdx.y;
  if (r >= rows) return;
  double2 acc = *reinterpret_cast<const double2 *>(x + r + ldx * j);
  for (int q = 1; q < nslab; ++q) {
    const double2 v = *reinterpret_cast<const double2 *>(x + slab_stride * q + r + ldx * j);
    acc.x += v.x; acc.y += v.y;
  }
  *reinterpret_cast<double2 *>(out + r + ldo * j) = acc;
}

int launch_sum_slabs(hipStream_t st, const double *x, int nslab, int64_t slab_stride, int rows,
                     int cols, int64_t ldx, double *out, int64_t ldo) {
  dim3 grid((rows / 2 + 255) / 256, cols);
  hipLaunchKernelGGL(sum_slabs_kernel, grid, dim3(256), 0, st, x, nslab, slab_stride, rows, ldx, out, ldo);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// Two blocked triangular solves, SUP rows per step: the diagonal step is a product with the
// inverted super-block (written to xtmp, then copied back over B's rows), followed by one
// rank-SUP update of the remaining rows.
int launch_potrs(hipStream_t st, const double *L, const double *inv, int n_pad, int batch,
                 double *B, int64_t ldb, int64_t strideB, int nrhs_pad, double *xtmp) {
  const int nb = n_pad / TILE, nsup = inv_nsup(n_pad);
  const int64_t sS = (int64_t)n_pad * n_pad, sI = inv_group_elems(n_pad);
  const int64_t sX = (int64_t)SUP * nrhs_pad;
  auto diag_step = [&](int c, int bs, bool transposed) -> int {
    if (batch == 1 && bs > TILE) {
      // split-K: the product with the inverted super-block is only bs/128 x nrhs/128 tiles of
      // depth bs -- too few and too deep to fill the chip -- so each 128-deep slice of K goes
      // to its own slab of xtmp (the GEMM's batch index) and a small kernel sums the slabs
      // back over B's rows
      const int nsl = bs / TILE;
      GemmArgs d{};
      d.A = inv + (int64_t)c * SUP * SUP; d.lda = SUP;
      d.strideA = transposed ? (int64_t)TILE : (int64_t)TILE * SUP;
      d.B = B + (int64_t)c * SUP; d.ldb = ldb; d.strideB = TILE;
      d.C = xtmp; d.ldc = SUP; d.strideC = sX;
      d.M = bs; d.N = nrhs_pad; d.K = TILE; d.batch = nsl; d.alpha = 1.0; d.beta = 0.0;
      GP_TRY(launch_gemm(st, transposed, false, d));
      dim3 grid((bs / 2 + 255) / 256, nrhs_pad);
      hipLaunchKernelGGL(sum_slabs_kernel, grid, dim3(256), 0, st, xtmp, nsl, sX, bs, (int64_t)SUP,
                         B + (int64_t)c * SUP, ldb);
      GP_HIP(hipGetLastError());
      return GPIRT_OK;
    }
    GemmArgs d{};
    d.A = inv + (int64_t)c * SUP * SUP; d.lda = SUP; d.strideA = sI;
    d.B = B + (int64_t)c * SUP; d.ldb = ldb; d.strideB = strideB;
    d.C = xtmp; d.ldc = SUP; d.strideC = sX;
    d.M = bs; d.N = nrhs_pad; d.K = bs; d.batch = batch; d.alpha = 1.0; d.beta = 0.0;
    d.k_trim = transposed ? 2 : 1;
    GP_TRY(launch_gemm(st, transposed, false, d));
    for (int z = 0; z < batch; ++z)
      GP_HIP(hipMemcpy2DAsync(B + strideB * z + (int64_t)c * SUP, sizeof(double) * ldb, xtmp + sX * z,
                              sizeof(double) * SUP, sizeof(double) * bs, nrhs_pad,
                              hipMemcpyDeviceToDevice, st));
    return GPIRT_OK;
  };
  for (int c = 0; c < nsup; ++c) {          // L W = B
    const int bs = min(4, nb - 4 * c) * TILE, r0 = c * SUP, below = n_pad - r0 - bs;
    GP_TRY(diag_step(c, bs, false));
    if (below <= 0) break;
    GemmArgs u{};
    u.A = L + (r0 + bs) + (int64_t)n_pad * r0; u.lda = n_pad; u.strideA = sS;
    u.B = B + r0; u.ldb = ldb; u.strideB = strideB;
    u.C = B + r0 + bs; u.ldc = ldb; u.strideC = strideB;
    u.M = below; u.N = nrhs_pad; u.K = bs; u.batch = batch; u.alpha = -1.0; u.beta = 1.0;
    GP_TRY(launch_gemm(st, false, false, u));
  }
  for (int c = nsup - 1; c >= 0; --c) {     // L' X = W
    const int bs = min(4, nb - 4 * c) * TILE, r0 = c * SUP;
    GP_TRY(diag_step(c, bs, true));
    if (!c) break;
    GemmArgs u{};  // B_above -= L[block c, above]^T B_c
    u.A = L + r0; u.lda = n_pad; u.strideA = sS;
    u.B = B + r0; u.ldb = ldb; u.strideB = strideB;
    u.C = B; u.ldc = ldb; u.strideC = strideB;
    u.M = r0; u.N = nrhs_pad; u.K = bs; u.batch = batch; u.alpha = -1.0; u.beta = 1.0;
    GP_TRY(launch_gemm(st, true, false, u));
  }
  return GPIRT_OK;
}

}  // namespace gpirt
