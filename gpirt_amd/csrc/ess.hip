// Per-unit kernels of the sweep: the ordinal-probit log-likelihood reductions and the
// elliptical-slice updates of each item's latent response function f and of its
// (beta, threshold) parameters.  A "unit" is one (item, period) pair (one item when the
// response function is shared by all periods); one 64-lane wave owns one unit for the whole
// shrinking-bracket loop, so units never synchronise with each other.
//
//   log-likelihood    utils/log_like_fn.m:3-4, link utils/sigmoid_fn.m:4,
//                     ordinal form gpirt-sythetic.R:159-167
//   slice update      utils/elliptical_slice.m:47-102, utils/sample_fs.m:1-6
#include "kernels.h"
#include "phi_fast.h"

namespace gpirt {

constexpr int WPB = 4;  // waves (units) per workgroup
constexpr int WMAX = 8; // most waves a unit can have
// waves of a workgroup whose units have WPU waves each
constexpr int block_waves(int WPU) { return WPU > WPB ? WPU : WPB; }

__device__ inline double bcast0(double v) { return __shfl(v, 0, 64); }

// The cell of response category c (1..C) lies between thresholds c - 1 and c.  The row loops read both
// ends with one 16-byte load from a table of pairs indexed by the response itself; entry 0, a missing
// response, is (-inf, +inf), whose cell has probability exactly 1 -- missing responses need no branch
// and no select.  Called by every lane of a wave with `own` = threshold number `lane` (-inf for lane 0,
// +inf for lane C); the lower end comes from the lane below by a shuffle (a read of what another lane
// has just written to LDS needs a fence: to the compiler the two addresses are unrelated).
__device__ inline void threshold_pairs(double own, double *sb, double2 *sp, int C, int lane) {
  const double below = __shfl_up(own, 1, 64);
  if (lane <= C) {
    sb[lane] = own;
    sp[lane] = lane == 0 ? make_double2(-INFINITY, INFINITY) : make_double2(below, own);
  }
}

// ---------------------------------------------------------------------------------------
// plain reductions (parity entry point gpirt_ordinal_loglik, end-of-sweep log-likelihood)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * WPB) void loglik_unit_kernel(
    const double *__restrict__ g, int64_t ldg, const int8_t *__restrict__ y, int64_t ldy,
    const double *__restrict__ bthr, int n, int U, int C, double *__restrict__ ll_unit) {
  __shared__ double sb[WPB][MAXC + 2];
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  phi_table_to_lds(sphi);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u = blockIdx.x * WPB + w;
  if (u >= U) return;
  if (lane <= C) sb[w][lane] = lane == 0 ? -INFINITY : (lane == C ? INFINITY : bthr[lane + (int64_t)(C + 1) * u]);
  LogProduct prod;
  prod.reset();
  for (int r = lane; r < n; r += 64) {
    int yy = y[r + ldy * u];
    if (yy > 0) {
      double gg = g[r + ldg * u];
      prod.mul(ordinal_cell_prob(sb[w][yy - 1] - gg, sb[w][yy] - gg, sphi));
    }
  }
  double acc = wave_sum(prod.log_value());
  if (lane == 0) ll_unit[u] = acc;
}

// per-row sums: thread per row, loop over units (rows contiguous across lanes)
__global__ void loglik_row_kernel(const double *__restrict__ g, int64_t ldg,
                                  const int8_t *__restrict__ y, int64_t ldy,
                                  const double *__restrict__ bthr, int n, int U, int C,
                                  double *__restrict__ ll_row) {
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  phi_table_to_lds(sphi);
  __syncthreads();
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  LogProduct prod;
  prod.reset();
  for (int u = 0; u < U; ++u) {
    int yy = y[r + ldy * u];
    if (yy > 0) {
      const double *b = bthr + (int64_t)(C + 1) * u;
      double lo = yy == 1 ? -INFINITY : b[yy - 1], hi = yy == C ? INFINITY : b[yy];
      double gg = g[r + ldg * u];
      prod.mul(ordinal_cell_prob(lo - gg, hi - gg, sphi));
    }
  }
  ll_row[r] = prod.log_value();
}

int launch_loglik(hipStream_t st, const double *g, int64_t ldg, const int8_t *y, int64_t ldy,
                  const double *bthr, int n, int U, int C, double *ll_unit, double *ll_row) {
  if (ll_unit) {
    hipLaunchKernelGGL(loglik_unit_kernel, dim3((U + WPB - 1) / WPB), dim3(64 * WPB), 0, st, g, ldg,
                       y, ldy, bthr, n, U, C, ll_unit);
    GP_HIP(hipGetLastError());
  }
  if (ll_row) {
    hipLaunchKernelGGL(loglik_row_kernel, dim3((n + 255) / 256), dim3(256), 0, st, g, ldg, y, ldy,
                       bthr, n, U, C, ll_row);
    GP_HIP(hipGetLastError());
  }
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// standard normals for every (row, unit): Z[r + ld*u], keyed by the GLOBAL unit id so that
// the stream does not depend on how items are sharded; zero in the pad.
// ---------------------------------------------------------------------------------------
__global__ void fill_normal_kernel(double *__restrict__ Z, int64_t ld, int Np, int Np_pad,
                                   int U_pad, int mloc, int ML, int m, int j0, uint64_t seed,
                                   uint32_t sweep) {
  int q = blockIdx.x * blockDim.x + threadIdx.x;  // pair index along rows
  int u = blockIdx.y;
  if (2 * q >= Np_pad) return;
  double z0 = 0.0, z1 = 0.0;
  if (u % ML < mloc && 2 * q < Np) {
    uint32_t ug = (uint32_t)((j0 + u % ML) + m * (u / ML));
    rng_normal_pair(seed, TAG_F_Z, sweep, ug, (uint32_t)q, z0, z1);
    if (2 * q + 1 >= Np) z1 = 0.0;
  }
  *reinterpret_cast<double2 *>(Z + 2 * q + ld * (int64_t)u) = make_double2(z0, z1);
}

int launch_fill_normal(hipStream_t st, double *Z, int64_t ld, int Np, int Np_pad, int U_pad,
                       int mloc, int ML, int m, int j0, uint64_t seed, uint32_t sweep) {
  dim3 block(256), grid((Np_pad / 2 + 255) / 256, U_pad);
  hipLaunchKernelGGL(fill_normal_kernel, grid, block, 0, st, Z, ld, Np, Np_pad, U_pad, mloc, ML, m,
                     j0, seed, sweep);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// `count` draws of stream (seed, tag, sweep, unit): the parity entry points gpirt_rng_*
__global__ void rng_fill_kernel(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                                int64_t count, int normal, double *__restrict__ out) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (2 * q >= count) return;
  double a, b;
  if (normal) rng_normal_pair(seed, tag, sweep, unit, (uint32_t)q, a, b);
  else rng_uniform_pair(seed, tag, sweep, unit, (uint32_t)q, a, b);
  out[2 * q] = a;
  if (2 * q + 1 < count) out[2 * q + 1] = b;
}

int launch_rng_fill(hipStream_t st, uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                    int64_t count, int normal, double *out) {
  int64_t pairs = (count + 1) / 2;
  if (pairs == 0) return GPIRT_OK;
  hipLaunchKernelGGL(rng_fill_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, st, seed,
                     tag, sweep, unit, count, normal, out);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// A unit is owned either by one wave (WPU = 1, four units per workgroup) or, when it has
// many rows, by the four waves of a workgroup (WPU = 4): units need different numbers of
// likelihood evaluations, and with one workgroup per unit the dispatcher evens that out
// instead of leaving the longest unit of each SIMD as a tail.
// ---------------------------------------------------------------------------------------
template <int WPU>
__device__ inline double unit_sum(double acc, double *red, int w) {
  const double v = wave_sum(acc);
  if (WPU == 1) return bcast0(v);
  __syncthreads();                       // everyone is done with the previous sums
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  const double lo4 = (red[0] + red[1]) + (red[2] + red[3]);   // fixed order: identical in every wave
  if (WPU == 4) return lo4;
  return lo4 + ((red[4] + red[5]) + (red[6] + red[7]));
}

// A proposal is accepted when its log-likelihood exceeds the slice height hh
// (elliptical_slice.m:87).  Every cell adds log p <= 0, so the sum over the rows done so far
// bounds the final sum from above, and the running product's exponents bound that in turn
// (mantissas lie in [1/2, 1)):  sum so far < ln2 * (sum of the lanes' exponents).  Once this
// integer bound is at or below hh the proposal is rejected whatever the remaining rows hold,
// and the pass stops there: same decision, most of a far-off proposal's rows never evaluated.
// Checked every SLICE_CHECK iterations (8: with 4 the waves of a unit meet at a barrier every four cells a lane,
// 1.42 + 1.42 ms for the two updates against 1.39 + 1.34; 16: 1.39 + 1.36); uniform over the unit's waves.  The waves exchange
// their sums through one of two slots, alternating from check to check: a wave can only write
// a slot again after the barrier of the check in between, which every wave passes after its
// reads of that slot.
#ifndef GPIRT_SLICE_CHECK
#define GPIRT_SLICE_CHECK 8
#endif
constexpr int SLICE_CHECK = GPIRT_SLICE_CHECK;
template <int WPU>
__device__ inline bool surely_rejected(int e, double hh, int *redi, int w) {
  int v = e;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  if (WPU != 1) {
    if ((threadIdx.x & 63) == 0) redi[w] = v;
    __syncthreads();
    v = (redi[0] + redi[1]) + (redi[2] + redi[3]);
    if (WPU == 8) v += (redi[4] + redi[5]) + (redi[6] + redi[7]);
  }
  return (double)v * 0.69314718055994530942 <= hh;
}

// ---------------------------------------------------------------------------------------
// slice update of f (a4 + a5)
// ---------------------------------------------------------------------------------------
// Element r of a column whose base address is uniform over the wave, through a 32-bit byte
// offset: the load then takes the base from scalar registers and one vector register of offset
// (a 64-bit index costs four extra vector instructions per load).
__device__ inline double col_at(const double *base, uint32_t r) {
#ifdef GPIRT_PROBE_SHORT_COLUMN
  r &= 1023;                                           // timing probe: every pass hits the cache (wrong numbers)
#endif
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(base) + (r << 3));
}
__device__ inline int col_at(const int8_t *base, uint32_t r) {
#ifdef GPIRT_PROBE_SHORT_COLUMN
  r &= 1023;
#endif
  return *reinterpret_cast<const int8_t *>(reinterpret_cast<const char *>(base) + r);
}

// Software-pipelined and branch-free: the loads of the next row are in flight while the
// current cell is evaluated (a wave that waits for y before it asks for f pays two memory
// round trips per row, and four waves per SIMD do not hide that).  Two rows per trip, each with
// its own set of registers, so that nothing is copied from "next" to "current".  Indices past
// the end are clamped to the last row and their cells masked out.
// `xm` holds the mean itself (b0 = 0, b1 = 1, b2 = 0) or the inputs of the quadratic mean.
template <int WPU>
__device__ inline double ll_f_pass(const double *__restrict__ f, const double *__restrict__ nu,
                                   const double *__restrict__ xm, double b0, double b1, double b2,
                                   const int8_t *__restrict__ y, const double2 *sp, const double *sphi,
                                   int n, int row0, double c, double s, double *red, int w, double hh,
                                   int (*redi)[WMAX], int &cp) {
  LogProduct prod;
  prod.reset();
  const uint32_t stride = 64 * WPU, last = (uint32_t)(n - 1);
  const int iters = (n + (int)stride - 1) / (int)stride;   // uniform over the unit
  auto cell = [&](int yy, double fr, double nr, double xr, bool live) {
    const double gg = fma(fma(b2, xr, b1), xr, b0) + (fr * c + nr * s);
    const double2 b = sp[live ? yy : 0];
    prod.mul(ordinal_cell_prob(b.x - gg, b.y - gg, sphi));
  };
  uint32_t r = (uint32_t)row0;
  const uint32_t r0 = min(r, last);
  int ya = col_at(y, r0);
  double fa = col_at(f, r0), na = col_at(nu, r0), xa = col_at(xm, r0);
  // trips whose two rows exist for every lane of the unit: no cell is masked by its row number
  const int whole = n / (int)stride;
  int it = 0;
  for (; it + 2 <= whole; it += 2, r += 2 * stride) {
    const uint32_t r1 = min(r + stride, last);
    const int yb = col_at(y, r1);
    const double fb = col_at(f, r1), nb = col_at(nu, r1), xb = col_at(xm, r1);
    cell(ya, fa, na, xa, true);
    const uint32_t r2 = min(r + 2 * stride, last);
    ya = col_at(y, r2);
    fa = col_at(f, r2); na = col_at(nu, r2); xa = col_at(xm, r2);
    cell(yb, fb, nb, xb, true);
    if ((it & (SLICE_CHECK - 1)) == SLICE_CHECK - 2 && it + 2 < iters && hh > -INFINITY &&
        surely_rejected<WPU>(prod.e, hh, redi[cp++ & 1], w))
      return -INFINITY;
  }
  if (it < iters) {                      // the last trip: rows past the end are masked out
    const uint32_t r1 = min(r + stride, last);
    const int yb = col_at(y, r1);
    const double fb = col_at(f, r1), nb = col_at(nu, r1), xb = col_at(xm, r1);
    cell(ya, fa, na, xa, r <= last);
    cell(yb, fb, nb, xb, r + stride <= last);
  }
  return unit_sum<WPU>(prod.log_value(), red, w);
}

template <int WPU>
__global__ __launch_bounds__(64 * block_waves(WPU)) void ess_f_kernel(EssFArgs a) {
  __shared__ double sbb[WPB][MAXC + 2];
  __shared__ __attribute__((aligned(16))) double2 spp[WPB][MAXC + 2];
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  __shared__ double red[WMAX];
  __shared__ int redi[2][WMAX];
  phi_table_to_lds(sphi);
  __syncthreads();
  int cp = 0;   // checks so far: consecutive checks alternate between the two exchange slots
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  // the unit is the same for every lane of a wave: say so, and its columns' addresses stay scalar
  const int u = WPU == 1 ? __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + w) : blockIdx.x + a.u0;
  if (u >= a.U_pad || u % a.ML >= a.mloc) return;    // uniform over the workgroup when WPU = 4
  double *sb = sbb[WPU == 1 ? w : 0];
  double2 *sp = spp[WPU == 1 ? w : 0];
  const int C = a.C, n = a.Np;
  const int row0 = WPU == 1 ? lane : lane + 64 * w;
  if (WPU == 1 || w == 0) {
    const double own = lane == 0 ? -INFINITY : (lane < C ? a.bthr[lane + (int64_t)(C + 1) * u] : INFINITY);
    threshold_pairs(own, sb, sp, C, lane);
  }
  if (WPU != 1) __syncthreads();
  double *f = a.f + a.ld * u;
  const double *nu = a.nu + a.ld * u;
  const double *mu = a.mu ? a.mu + a.ld_mu * u : nullptr;
  const int8_t *y = a.y + a.ldy * u;
  const double *x = a.x ? a.x + a.x_stride * (u / a.ML) : nullptr;
  double b0 = 0, b1 = 0, b2 = 0;
  if (a.beta) { b0 = a.beta[3 * u]; b1 = a.beta[3 * u + 1]; b2 = a.beta[3 * u + 2]; }
  // a mean given row by row goes through the same expression: 0 + 1 * mu + 0 * mu^2 = mu exactly
  const double *xm = mu ? mu : x;
  if (mu) { b0 = 0.0; b1 = 1.0; b2 = 0.0; }
  const uint32_t ug = (uint32_t)((a.j0 + u % a.ML) + a.m * (u / a.ML));
  const double *un = a.unif ? a.unif + (int64_t)(a.R + 2) * u : nullptr;
  auto draw = [&](int d) -> double {
    return un ? un[d] : rng_uniform_at(a.seed, TAG_F_U, a.sweep, ug, (uint32_t)d);
  };

  const double cur = ll_f_pass<WPU>(f, nu, xm, b0, b1, b2, y, sp, sphi, n, row0, 1.0, 0.0, red, w,
                                    -INFINITY, redi, cp);
  const double hh = log(draw(0)) + cur;              // elliptical_slice.m:66
  double phi = draw(1) * kTwoPi;                     // :72
  double pmin = phi - kTwoPi, pmax = phi;            // :73-74
  int r = 0;
  bool ok = false;
  double l = cur, c = 1.0, s = 0.0;
  while (r < a.R) {
    sincos(phi, &s, &c);
    l = ll_f_pass<WPU>(f, nu, xm, b0, b1, b2, y, sp, sphi, n, row0, c, s, red, w, hh, redi, cp);  // :85-86
    ++r;
    if (l > hh) { ok = true; break; }                // :87
    if (phi > 0.0) pmax = phi;                       // :92-96
    else if (phi < 0.0) pmin = phi;
    else break;
    phi = draw(1 + r) * (pmax - pmin) + pmin;        // :100
  }
  if (ok)
    for (int i = row0; i < n; i += 64 * WPU) f[i] = f[i] * c + nu[i] * s;
  if (lane == 0 && (WPU == 1 || w == 0)) {
    if (a.rounds) a.rounds[u] = ok ? r : -r;
    if (a.ll) a.ll[u] = ok ? l : cur;
    if (a.counters) {
      atomicAdd(&a.counters[0], (unsigned long long)(1 + r));
      if (!ok) atomicAdd(&a.counters[1], 1ull);
    }
  }
}

// rows per unit from which a unit gets a whole workgroup
constexpr int WIDE_UNIT_ROWS = 1024;

int launch_ess_f(hipStream_t st, const EssFArgs &a) {
  // Eight waves (EssFArgs::wide8, the caller's choice: it changes the order of a unit's sums): with few units
  // the launch is as long as its longest unit, and a wave that has its SIMD to itself issues at half the rate of
  // two -- n = 10 000: 512 units 0.37 -> 0.30 ms, 1024 units 0.48 -> 0.46, 2048 and more: no shorter.  (The
  // parameter update needs 151 registers: eight-wave workgroups then fit one to a CU, and it is slower at every size.)
  if (a.wide8 && a.Np >= WIDE_UNIT_ROWS)
    hipLaunchKernelGGL(ess_f_kernel<8>, dim3(a.U_pad), dim3(64 * 8), 0, st, a);
  else if (a.Np >= WIDE_UNIT_ROWS)
    hipLaunchKernelGGL(ess_f_kernel<4>, dim3(a.ucount ? a.ucount : a.U_pad), dim3(64 * WPB), 0, st, a);
  else
    hipLaunchKernelGGL(ess_f_kernel<1>, dim3((a.U_pad + WPB - 1) / WPB), dim3(64 * WPB), 0, st, a);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// joint slice update of (beta, thresholds) of each unit (a8 + the IRF mean part).  The
// vector is e[3] (whitened coefficients) and delta[C-1] (b_1 = delta_1,
// b_c = b_{c-1} + exp(delta_c)), prior N(0,I); a switched-off coefficient and fixed
// thresholds do not move.
// ---------------------------------------------------------------------------------------
// The vector lives one element per lane (lane k < C + 2 owns element k): nothing of it occupies
// registers during the row loop, and the exponentials of a proposal are one call, not C - 2.
__device__ inline double lane_value(double v, int k) { return __shfl(v, k, 64); }

// (beta, thresholds) of the vector `pv`; lanes 0..2 hold the coefficients' prior mean and sd in
// pm_k, ps_k.  The thresholds accumulate left to right, every lane doing the same additions.
__device__ inline void params_from_white(double pv, double pm_k, double ps_k, int C, int lane,
                                         double &b0, double &b1, double &b2, double *sb, double2 *sp) {
  const double bk = pm_k + ps_k * pv;
  b0 = lane_value(bk, 0);
  b1 = lane_value(bk, 1);
  b2 = lane_value(bk, 2);
  const double ek = exp(pv);
  double acc = lane_value(pv, 3), mine = acc;
  for (int c = 2; c < C; ++c) {
    acc += lane_value(ek, c + 2);
    if (lane == c) mine = acc;
  }
  threshold_pairs(lane == 0 ? -INFINITY : (lane < C ? mine : INFINITY), sb, sp, C, lane);
}

template <int WPU>
__device__ inline double ll_param_pass(const double *__restrict__ f, const double *__restrict__ x,
                                       double b0, double b1, double b2,
                                       const int8_t *__restrict__ y, const double2 *sp,
                                       const double *sphi, int n, int row0, double *red, int w,
                                       double hh, int (*redi)[WMAX], int &cp) {
  LogProduct prod;                       // pipelined like ll_f_pass, same early rejection
  prod.reset();
  const uint32_t stride = 64 * WPU, last = (uint32_t)(n - 1);
  const int iters = (n + (int)stride - 1) / (int)stride;
  auto cell = [&](int yy, double fr, double xr, bool live) {
    const double gg = fma(fma(b2, xr, b1), xr, b0) + fr;
    const double2 b = sp[live ? yy : 0];
    prod.mul(ordinal_cell_prob(b.x - gg, b.y - gg, sphi));
  };
  uint32_t r = (uint32_t)row0;
  const uint32_t r0 = min(r, last);
  int ya = col_at(y, r0);
  double fa = col_at(f, r0), xa = col_at(x, r0);
  const int whole = n / (int)stride;     // as in ll_f_pass
  int it = 0;
  for (; it + 2 <= whole; it += 2, r += 2 * stride) {
    const uint32_t r1 = min(r + stride, last);
    const int yb = col_at(y, r1);
    const double fb = col_at(f, r1), xb = col_at(x, r1);
    cell(ya, fa, xa, true);
    const uint32_t r2 = min(r + 2 * stride, last);
    ya = col_at(y, r2);
    fa = col_at(f, r2); xa = col_at(x, r2);
    cell(yb, fb, xb, true);
    if ((it & (SLICE_CHECK - 1)) == SLICE_CHECK - 2 && it + 2 < iters && hh > -INFINITY &&
        surely_rejected<WPU>(prod.e, hh, redi[cp++ & 1], w))
      return -INFINITY;
  }
  if (it < iters) {
    const uint32_t r1 = min(r + stride, last);
    const int yb = col_at(y, r1);
    const double fb = col_at(f, r1), xb = col_at(x, r1);
    cell(ya, fa, xa, r <= last);
    cell(yb, fb, xb, r + stride <= last);
  }
  return unit_sum<WPU>(prod.log_value(), red, w);
}

template <int WPU>
__global__ __launch_bounds__(64 * block_waves(WPU)) void ess_param_kernel(EssParamArgs a) {
  __shared__ double sbb[WPB][MAXC + 2];
  __shared__ __attribute__((aligned(16))) double2 spp[WPB][MAXC + 2];
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  __shared__ double red[WMAX];
  __shared__ int redi[2][WMAX];
  phi_table_to_lds(sphi);
  __syncthreads();
  int cp = 0;   // checks so far: consecutive checks alternate between the two exchange slots
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int u = WPU == 1 ? __builtin_amdgcn_readfirstlane(blockIdx.x * WPB + w) : blockIdx.x + a.u0;
  if (u >= a.U_pad || u % a.ML >= a.mloc) return;    // uniform over the workgroup when WPU = 4
  double *sb = sbb[WPU == 1 ? w : 0];
  double2 *sp = spp[WPU == 1 ? w : 0];
  const int row0 = WPU == 1 ? lane : lane + 64 * w;
  const int C = a.C, n = a.Np, D = C + 2, jl = u % a.ML;
  const double *f = a.f + a.ld * u;
  const int8_t *y = a.y + a.ldy * u;
  const double *x = a.x + a.x_stride * (u / a.ML);
  const uint32_t ug = (uint32_t)((a.j0 + jl) + a.m * (u / a.ML));

  // this lane's element of the vector: e[0..2] on lanes 0..2, delta[0..C-2] on lanes 3..C+1
  const bool is_beta = lane < 3, mine = lane < D;
  double pm_k = 0.0, ps_k = 0.0;
  if (is_beta) { pm_k = a.bpm[lane + 3 * jl]; ps_k = a.bps[lane + 3 * jl]; }
  const bool is_free = mine && (is_beta ? ps_k > 0.0 : !a.fixed_thr);
  const bool rotates = is_beta || !a.fixed_thr;
  double v = 0.0;
  if (mine && !is_beta) v = a.delta[(lane - 3) + (int64_t)(C - 1) * u];
  if (is_beta && is_free) v = a.ebeta[lane + 3 * u];
  const double nv = is_free ? rng_normal_at(a.seed, TAG_P_Z, a.sweep, ug, (uint32_t)lane) : 0.0;
  const bool anyfree = __any(is_free);
  double pv = v;

  double b0, b1, b2;
  params_from_white(v, pm_k, ps_k, C, lane, b0, b1, b2, sb, sp);
  if (WPU != 1) __syncthreads();          // the thresholds every wave just wrote (same values)
  // the slice update of f that ran just before left the log-likelihood of exactly this state
  const double cur = a.ll_in ? a.ll_in[u]
                             : ll_param_pass<WPU>(f, x, b0, b1, b2, y, sp, sphi, n, row0, red, w, -INFINITY, redi, cp);
  int r = 0;
  bool ok = false;
  double l = cur;
  if (anyfree) {
    const double hh = log(rng_uniform_at(a.seed, TAG_P_U, a.sweep, ug, 0)) + cur;
    double phi = rng_uniform_at(a.seed, TAG_P_U, a.sweep, ug, 1) * kTwoPi;
    double pmin = phi - kTwoPi, pmax = phi;
    while (r < a.R) {
      double s, c;
      sincos(phi, &s, &c);
      pv = rotates ? v * c + nv * s : v;
      // (every wave has left the last pass: its final sum, or the check that ended it, is a barrier)
      params_from_white(pv, pm_k, ps_k, C, lane, b0, b1, b2, sb, sp);
      if (WPU != 1) __syncthreads();
      l = ll_param_pass<WPU>(f, x, b0, b1, b2, y, sp, sphi, n, row0, red, w, hh, redi, cp);
      ++r;
      if (l > hh) { ok = true; break; }
      if (phi > 0.0) pmax = phi;
      else if (phi < 0.0) pmin = phi;
      else break;
      phi = rng_uniform_at(a.seed, TAG_P_U, a.sweep, ug, (uint32_t)(1 + r)) * (pmax - pmin) + pmin;
    }
  }
  if (!ok) {  // keep the current parameters
    params_from_white(v, pm_k, ps_k, C, lane, b0, b1, b2, sb, sp);
    l = cur;
  }
  if (WPU != 1) __syncthreads();
  const double fin = ok ? pv : v;
  if (WPU == 1 || w == 0) {
    if (is_beta) a.ebeta[lane + 3 * u] = fin;
    else if (mine) a.delta[(lane - 3) + (int64_t)(C - 1) * u] = fin;
    if (lane == 0) {
      a.beta[3 * u] = b0; a.beta[3 * u + 1] = b1; a.beta[3 * u + 2] = b2;
      a.betaS[u] = b0; a.betaS[u + a.U_pad] = b1; a.betaS[u + 2 * (int64_t)a.U_pad] = b2;
      if (a.ll) a.ll[u] = l;
      if (a.counters && anyfree) {
        atomicAdd(&a.counters[0], (unsigned long long)(1 + r));
        if (!ok) atomicAdd(&a.counters[1], 1ull);
      }
    }
    if (lane <= C) {
      double bv = sb[lane];
      a.bthr[lane + (int64_t)(C + 1) * u] = bv;
      a.bthrS[u + (int64_t)a.U_pad * lane] = bv;
    }
  }
}

int launch_ess_param(hipStream_t st, const EssParamArgs &a) {
  if (a.Np >= WIDE_UNIT_ROWS)
    hipLaunchKernelGGL(ess_param_kernel<4>, dim3(a.ucount ? a.ucount : a.U_pad), dim3(64 * WPB), 0, st, a);
  else
    hipLaunchKernelGGL(ess_param_kernel<1>, dim3((a.U_pad + WPB - 1) / WPB), dim3(64 * WPB), 0, st, a);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

}  // namespace gpirt
