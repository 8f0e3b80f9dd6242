// Respondent step and the state snap.
//
// Each respondent's trajectory theta_i[1..H] gets one elliptical-slice update
// (utils/elliptical_slice.m:47-102) under the temporal GP prior, with the likelihood read
// off the grid: f* is looked up at the nearest grid point (the index rule of main.m:69; on request
// at the grid point below, the rule of gpirt-sythetic.R:190) and
// the ordinal-probit cell of log_like_fn.m:3-4 / gpirt-sythetic.R:159-167 is summed over
// this rank's items.  All respondents advance in lock-step rounds so that, when items are
// sharded over GPUs, one n-vector sum-all-reduce per round is the only exchange.
#include <type_traits>
#include "kernels.h"
#include "phi_fast.h"

namespace gpirt {

// nearest grid point (main.m:69), or the grid point at or below x (gpirt-sythetic.R:190)
__device__ inline int grid_index(double x, double lo, double step, int G, int truncate) {
  const double t = (x - lo) / step;
  double r = truncate ? floor(t) : rint(t);
  if (!(r > 0.0)) return 0;
  if (r > (double)(G - 1)) return G - 1;
  return (int)r;
}

__device__ inline double prior_mean(const double *tpm, int i, int h) {
  return tpm[2 * i] + tpm[2 * i + 1] * (double)(h + 1);
}

// bucketed evaluation (H = 1): the kernel that fixes the theta to be evaluated next also
// files the respondent under that theta's grid index
__device__ inline void bucket_note(const ThetaArgs &a, int v, double th) {
  const int g = grid_index(th, a.grid_lo, a.grid_step, a.G, a.index_truncate);
  a.gidx[v] = g;
  atomicAdd(&a.ghist[g], 1);
}

// ellipse draw nu_i = L_t z + s0 z_H + s1 t z_{H+1}, centred state x_i = theta_i - mean_i
__global__ void theta_prepare_kernel(ThetaArgs a) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  const int H = a.H, n = a.n;
  double z[MAXH + 2];
  for (int k = 0; k < H + 2; ++k) z[k] = rng_normal_at(a.seed, TAG_T_Z, a.sweep, (uint32_t)i, (uint32_t)k);
  const double s0 = a.tps[2 * i], s1 = a.tps[2 * i + 1];
  for (int h = 0; h < H; ++h) {
    double t = (double)(h + 1);
    double acc = s0 * z[H] + s1 * t * z[H + 1];
    for (int k = 0; k <= h; ++k) acc += a.Lt[h + H * k] * z[k];
    a.nu[i + (int64_t)n * h] = acc;
    a.xc[i + (int64_t)n * h] = a.theta[i + (int64_t)n * h] - prior_mean(a.tpm, i, h);
  }
  a.done[i] = 1;
  if (a.part) bucket_note(a, i, a.theta[i]);
}

int launch_theta_prepare(hipStream_t st, const ThetaArgs &a) {
  hipLaunchKernelGGL(theta_prepare_kernel, dim3((a.n + 127) / 128), dim3(128), 0, st, a);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// local-items log-likelihood of every active respondent.  WPR = 1: one wave per respondent
// (four respondents per workgroup).  WPR = 4: the four waves of a workgroup share one
// respondent -- for the late rounds, when few respondents are still active and the time of a
// round is the latency of one respondent's loop rather than the chip's throughput.
template <int WPR>
__global__ __launch_bounds__(256) void theta_eval_kernel(ThetaArgs a, int initial) {
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  __shared__ double red[4];
  phi_table_to_lds(sphi);
  __syncthreads();
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int v = WPR == 1 ? blockIdx.x * 4 + w : blockIdx.x;      // entry k*n + i
  const int count = initial ? a.n : THETA_SPEC * a.n;
  if (v >= count) return;
  const int i = v % a.n;
  const int64_t V = (int64_t)THETA_SPEC * a.n;
  if (!initial && v / a.n >= a.done[i]) {    // uniform over the workgroup when WPR = 4
    if (lane == 0 && (WPR == 1 || w == 0)) a.buf[v] = 0.0;
    return;
  }
  const int H = a.H, n = a.n, mloc = a.mloc, ML = a.ML, U_pad = a.U_pad;
  const int jfirst = WPR == 1 ? lane : lane + 64 * w, jstep = 64 * WPR;
  const double *b0 = a.betaS, *b1 = a.betaS + U_pad, *b2 = a.betaS + 2 * (int64_t)U_pad;
  LogProduct prod;
  prod.reset();
  for (int h = 0; h < H; ++h) {
    double th = initial ? a.theta[i + (int64_t)n * h] : a.prop[v + V * h] + prior_mean(a.tpm, i, h);
    const int g = grid_index(th, a.grid_lo, a.grid_step, a.G, a.index_truncate);
    const double xg = a.grid_lo + a.grid_step * g;
    const int p = a.constant_irf ? 0 : h;
    const double *Gr = a.Gt + (int64_t)U_pad * g + (int64_t)ML * p;
    const int8_t *yr = a.yT + (int64_t)ML * H * i + (int64_t)ML * h;
    if (jfirst >= mloc) continue;
    // software-pipelined, branch-free: y runs two items ahead (the thresholds are gathered by
    // category), everything else one item ahead; indices past the end are clamped to the
    // last item and those cells masked
    const int jend = mloc - 1, up = ML * p;
    int y1 = yr[jfirst], y2 = yr[min(jfirst + jstep, jend)];
    int yc = max(y1, 1);
    double G1 = Gr[jfirst], c0 = b0[up + jfirst], c1 = b1[up + jfirst], c2 = b2[up + jfirst];
    double lo = a.bthrS[up + jfirst + (int64_t)U_pad * (yc - 1)], hi = a.bthrS[up + jfirst + (int64_t)U_pad * yc];
    for (int jl = jfirst; jl < mloc; jl += jstep) {
      const int jn = min(jl + jstep, jend), un = up + jn;
      const int y3 = yr[min(jl + 2 * jstep, jend)];
      const int ycn = max(y2, 1);
      const double Gn = Gr[jn], d0 = b0[un], d1 = b1[un], d2 = b2[un];
      const double lon = a.bthrS[un + (int64_t)U_pad * (ycn - 1)], hin = a.bthrS[un + (int64_t)U_pad * ycn];
      const double gg = G1 + c0 + c1 * xg + c2 * xg * xg;
      const double pc = ordinal_cell_prob(lo - gg, hi - gg, sphi);
      prod.mul(y1 > 0 ? pc : 1.0);
      y1 = y2; y2 = y3; G1 = Gn; c0 = d0; c1 = d1; c2 = d2; lo = lon; hi = hin;
    }
  }
  const double acc = wave_sum(prod.log_value());
  if (WPR == 1) {
    if (lane == 0) a.buf[v] = acc;
  } else {
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.buf[v] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// Table variant.  A respondent's log-likelihood depends on theta only through the grid index,
// so with many respondents the G x units x C distinct log cell probabilities are tabulated
// once per sweep (4 tails and 5 logs per grid point and unit instead of 2 tails per cell and
// round) and a likelihood evaluation becomes a sum of gathered table entries.
__global__ __launch_bounds__(256) void theta_table_kernel(ThetaArgs a) {
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  __shared__ __attribute__((aligned(16))) double slog[LOG_LDS_DOUBLES];
  phi_table_to_lds(sphi);
  log_table_to_lds(slog);
  __syncthreads();
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  const int U_pad = a.U_pad, C = a.C;
  if (u >= U_pad || u % a.ML >= a.mloc) return;
  const double c0 = a.betaS[u], c1 = a.betaS[u + U_pad], c2 = a.betaS[u + 2 * (int64_t)U_pad];
  const double e_first = phi_upper(1e30, sphi);            // lower end: -inf
  // a workgroup stages the tail table once and serves every gridDim.y-th grid point with it
  for (int g = blockIdx.y; g < a.G; g += gridDim.y) {
    const double xg = a.grid_lo + a.grid_step * g;
    const double F = a.Gt[u + (int64_t)U_pad * g] + c0 + c1 * xg + c2 * xg * xg;
    double zlo = -1e30, e1 = e_first;
    double *out = a.tab + u + (int64_t)U_pad * C * g;
    for (int c = 0; c < C; ++c) {
      const double zhi = c + 1 < C ? fmin(fmax(a.bthrS[u + (int64_t)U_pad * (c + 1)] - F, -1e30), 1e30) : 1e30;
      const double e2 = phi_upper(fabs(zhi), sphi);
      out[(int64_t)U_pad * c] = log_pos(ordinal_cell_from_tails(zlo, zhi, e1, e2), slog);
      zlo = zhi; e1 = e2;
    }
  }
}

int launch_theta_table(hipStream_t st, const ThetaArgs &a) {
  // enough workgroups for two rounds of the chip at four to a CU
  const int nbx = (a.U_pad + 255) / 256, gy = std::max(1, std::min(a.G, 2048 / nbx));
  hipLaunchKernelGGL(theta_table_kernel, dim3(nbx, gy), dim3(256), 0, st, a);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

template <int WPR>
__global__ __launch_bounds__(256) void theta_eval_table_kernel(ThetaArgs a, int initial) {
  __shared__ double red[4];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int v = WPR == 1 ? blockIdx.x * 4 + w : blockIdx.x;      // entry k*n + i
  const int count = initial ? a.n : THETA_SPEC * a.n;
  if (v >= count) return;
  const int i = v % a.n;
  const int64_t V = (int64_t)THETA_SPEC * a.n;
  if (!initial && v / a.n >= a.done[i]) {    // uniform over the workgroup when WPR = 4
    if (lane == 0 && (WPR == 1 || w == 0)) a.buf[v] = 0.0;
    return;
  }
  const int H = a.H, n = a.n, mloc = a.mloc, ML = a.ML;
  const int64_t U_pad = a.U_pad;
  const int jfirst = WPR == 1 ? lane : lane + 64 * w, jstep = 64 * WPR;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  for (int h = 0; h < H; ++h) {
    double th = initial ? a.theta[i + (int64_t)n * h] : a.prop[v + V * h] + prior_mean(a.tpm, i, h);
    const int g = grid_index(th, a.grid_lo, a.grid_step, a.G, a.index_truncate);
    const int p = a.constant_irf ? 0 : h;
    const double *row = a.tab + U_pad * a.C * g + (int64_t)ML * p;
    const int8_t *yr = a.yT + (int64_t)ML * H * i + (int64_t)ML * h;
    int jl = jfirst;
    for (; jl + 3 * jstep < mloc; jl += 4 * jstep) {      // four independent gathers in flight
      const int y0 = yr[jl], y1 = yr[jl + jstep], y2 = yr[jl + 2 * jstep], y3 = yr[jl + 3 * jstep];
      const double v0 = row[jl + U_pad * (max(y0, 1) - 1)];
      const double v1 = row[jl + jstep + U_pad * (max(y1, 1) - 1)];
      const double v2 = row[jl + 2 * jstep + U_pad * (max(y2, 1) - 1)];
      const double v3 = row[jl + 3 * jstep + U_pad * (max(y3, 1) - 1)];
      s0 += y0 > 0 ? v0 : 0.0; s1 += y1 > 0 ? v1 : 0.0;
      s2 += y2 > 0 ? v2 : 0.0; s3 += y3 > 0 ? v3 : 0.0;
    }
    for (; jl < mloc; jl += jstep) {
      const int y0 = yr[jl];
      const double v0 = row[jl + U_pad * (max(y0, 1) - 1)];
      s0 += y0 > 0 ? v0 : 0.0;
    }
  }
  const double acc = wave_sum((s0 + s1) + (s2 + s3));
  if (WPR == 1) {
    if (lane == 0) a.buf[v] = acc;
  } else {
    if (lane == 0) red[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) a.buf[v] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// ---- bucketed evaluation (H = 1) -------------------------------------------------------
// With a workgroup per respondent every evaluation gathers 8 bytes out of each 40..64-byte
// piece of a 160 KB table row, and respondents at the same grid index re-read the same row
// from wherever the cache hierarchy still holds it: a round over all respondents moves
// ~1.6 GB.  Grouped by grid index, a row chunk is read once, kept in LDS, and serves every
// respondent of its bucket.
__global__ __launch_bounds__(1024) void theta_bucket_kernel(ThetaArgs a, int count) {
  __shared__ int hist[THETA_BUCKET_MAX_G];
  __shared__ int lane_tot[64];
  const int tid = threadIdx.x, n = count, G = a.G;
  for (int g = tid; g < THETA_BUCKET_MAX_G; g += 1024) {
    hist[g] = g < G ? a.ghist[g] : 0;
    if (g < G) a.ghist[g] = 0;                            // ready for the next round's counts
  }
  __syncthreads();
  // exclusive scan by the first wave: each lane owns a run of consecutive bins
  constexpr int RUN = THETA_BUCKET_MAX_G / 64;
  if (tid < 64) {
    int sum = 0;
    for (int k = 0; k < RUN; ++k) sum += hist[tid * RUN + k];
    int incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
      const int t = __shfl_up(incl, off, 64);
      if (tid >= off) incl += t;
    }
    int run = incl - sum;
    for (int k = 0; k < RUN; ++k) {
      const int c = hist[tid * RUN + k];
      hist[tid * RUN + k] = run;
      run += c;
    }
    lane_tot[tid] = incl;
  }
  __syncthreads();
  for (int g = tid; g < G; g += 1024) { a.bstart[g] = hist[g]; a.bcur[g] = hist[g]; }
  if (tid == 0) a.bstart[G] = lane_tot[63];
  (void)n;
}

// Every entry to a place in its bucket (the order inside a bucket is immaterial), one thread an entry over the
// whole chip; inside the single workgroup of the scan above this took three quarters of its 20 us.  The entries of a
// workgroup are counted per bucket in LDS first, so that a bucket's global cursor is advanced once per workgroup, not
// once per entry: a hundred and more entries share the buckets around the centre of the prior, and atomics on one
// address follow each other at the latency of the L2.
__global__ __launch_bounds__(1024) void theta_scatter_kernel(ThetaArgs a, int count) {
  __shared__ int cnt[THETA_BUCKET_MAX_G];
  __shared__ int base[THETA_BUCKET_MAX_G];
  const int tid = threadIdx.x, v = blockIdx.x * 1024 + tid;
  for (int g = tid; g < THETA_BUCKET_MAX_G; g += 1024) cnt[g] = 0;
  __syncthreads();
  const int g = v < count ? a.gidx[v] : -1;
  int rank = 0;
  if (g >= 0) rank = atomicAdd(&cnt[g], 1);
  __syncthreads();
  for (int b = tid; b < a.G; b += 1024)
    if (cnt[b] > 0) base[b] = atomicAdd(&a.bcur[b], cnt[b]);
  __syncthreads();
  if (g >= 0) a.order[base[g] + rank] = v;
}

constexpr int BUCKET_CHUNK_MAX = 1024;   // items per chunk: at most 16 per lane, held in registers

__global__ __launch_bounds__(256) void theta_eval_bucket_kernel(ThetaArgs a) {
  extern __shared__ double rows[];                        // [C][chunk]
  // Buckets are dealt out from the middle of the grid outwards: respondents crowd around the centre of
  // the prior, and the workgroups of the fullest buckets, which set the launch's length, start first.
  const int mid = a.G / 2, bx = blockIdx.x;
  const int g = (bx & 1) ? mid - (bx + 1) / 2 : (mid + bx / 2 < a.G ? mid + bx / 2 : a.G - 1 - bx);
  const int ch = blockIdx.y;
  const int b0 = a.bstart[g], b1 = a.bstart[g + 1];
  if (b0 >= b1) return;                                   // uniform over the workgroup
  const int chunk = a.chunk, C = a.C, j0 = ch * chunk, cnt = min(chunk, a.mloc - j0);
  const int64_t U_pad = a.U_pad;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  constexpr int PER = BUCKET_CHUNK_MAX / 64, STEP = 4;
  // a wave per respondent of the bucket.  Two respondents ahead: the respondent's number;
  // one ahead: the answers; both are in flight while the current table entries are summed.
  int r = b0 + w;
  int i0 = a.order[min(r, b1 - 1)], i1 = a.order[min(r + STEP, b1 - 1)];
  {  // stage the rows: all loads of a thread are issued before the first store
    const double *src = a.tab + U_pad * C * g + j0;
    constexpr int LOADS = 6144 / 256;                     // C * chunk <= 6144
    double tmp[LOADS];
    const int total = C * chunk;
    // (category, item) of entry e = thread + 256 q without a division per entry: the chunk is at least 64 wide
    int c = threadIdx.x / chunk, k = threadIdx.x - c * chunk;
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int e = threadIdx.x + 256 * q;
      tmp[q] = (e < total && k < cnt) ? src[U_pad * c + k] : 0.0;
      k += 256;
      while (k >= chunk) { k -= chunk; ++c; }
    }
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int e = threadIdx.x + 256 * q;
      if (e < total) rows[e] = tmp[q];
    }
  }
  int yv[PER], yn[PER];
  auto load_y = [&](int i, int *dst) {
    const int8_t *yr = a.yT + (int64_t)a.ML * (i % a.n) + j0;     // i: entry k*n + respondent
#pragma unroll
    for (int t = 0; t < PER; ++t) dst[t] = yr[min(lane + 64 * t, cnt - 1)];
  };
  load_y(i0, yv);
  __syncthreads();
  for (; r < b1; r += STEP) {
    const int i2 = a.order[min(r + 2 * STEP, b1 - 1)];
    load_y(i1, yn);
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int t = 0; t < PER; t += 2) {
      const int k0 = min(lane + 64 * t, cnt - 1), k1 = min(lane + 64 * (t + 1), cnt - 1);
      const double v0 = rows[(max(yv[t], 1) - 1) * chunk + k0];
      const double v1 = rows[(max(yv[t + 1], 1) - 1) * chunk + k1];
      s0 += (yv[t] > 0 && lane + 64 * t < cnt) ? v0 : 0.0;
      s1 += (yv[t + 1] > 0 && lane + 64 * (t + 1) < cnt) ? v1 : 0.0;
    }
    const double acc = wave_sum(s0 + s1);
    if (lane == 0) a.part[(int64_t)(a.slot0 + ch) * THETA_SPEC * a.n + i0] = acc;
#pragma unroll
    for (int t = 0; t < PER; ++t) yv[t] = yn[t];
    i0 = i1; i1 = i2;
  }
}

// The same evaluation for chunks of 512 or 1024 items, with the responses read from a copy in which the items a
// lane serves (lane, lane + 64, ...) lie side by side: one 8- or 16-byte load per respondent and lane instead of
// 8 or 16 byte loads with their address arithmetic.  The table rows sit behind a row of zeros, so that a response
// indexes its row directly and a missing one (0) adds zero: no select.  Same sums in the same order.
template <int PERC>
__global__ __launch_bounds__(256) void theta_eval_bucket_lane_kernel(ThetaArgs a) {
  extern __shared__ double rows[];                        // [C + 1][chunk], row 0 zero
  const int mid = a.G / 2, bx = blockIdx.x;
  const int g = (bx & 1) ? mid - (bx + 1) / 2 : (mid + bx / 2 < a.G ? mid + bx / 2 : a.G - 1 - bx);
  const int ch = blockIdx.y;
  const int b0 = a.bstart[g], b1 = a.bstart[g + 1];
  if (b0 >= b1) return;                                   // uniform over the workgroup
  constexpr int chunk = 64 * PERC, STEP = 4;
  const int C = a.C, j0 = ch * chunk, cnt = min(chunk, a.mloc - j0);
  const int64_t U_pad = a.U_pad;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  using word_t = typename std::conditional<PERC == 16, uint4, uint2>::type;
  auto load_y = [&](int i) -> word_t {
    return *reinterpret_cast<const word_t *>(a.yTl + (int64_t)a.ML * (i % a.n) + j0 + lane * PERC);
  };
  int r = b0 + w;
  int i0 = a.order[min(r, b1 - 1)], i1 = a.order[min(r + STEP, b1 - 1)];
  {  // stage the rows: all loads of a thread are issued before the first store
    const double *src = a.tab + U_pad * C * g + j0;
    constexpr int LOADS = 6144 / 256;                     // C * chunk <= 6144
    double tmp[LOADS];
    const int total = C * chunk;
    int c = threadIdx.x / chunk, k = threadIdx.x - c * chunk;
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int e = threadIdx.x + 256 * q;
      tmp[q] = (e < total && k < cnt) ? src[U_pad * c + k] : 0.0;
      k += 256;
      while (k >= chunk) { k -= chunk; ++c; }
    }
    for (int e = threadIdx.x; e < chunk; e += 256) rows[e] = 0.0;
#pragma unroll
    for (int q = 0; q < LOADS; ++q) {
      const int e = threadIdx.x + 256 * q;
      if (e < total) rows[chunk + e] = tmp[q];
    }
  }
  word_t yv = load_y(i0);
  __syncthreads();
  const double *mine = rows + lane;
  for (; r < b1; r += STEP) {
    const int i2 = a.order[min(r + 2 * STEP, b1 - 1)];
    const word_t yn = load_y(i1);
    uint32_t wd[4] = {yv.x, yv.y, 0u, 0u};
    if constexpr (PERC == 16) { wd[2] = yv.z; wd[3] = yv.w; }
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int t = 0; t < PERC; t += 2) {
      const uint32_t ya = (wd[t >> 2] >> (8 * (t & 3))) & 0xFFu, yb = (wd[(t + 1) >> 2] >> (8 * ((t + 1) & 3))) & 0xFFu;
      s0 += mine[ya * chunk + 64 * t];
      s1 += mine[yb * chunk + 64 * (t + 1)];
    }
    const double acc = wave_sum(s0 + s1);
    if (lane == 0) a.part[(int64_t)(a.slot0 + ch) * THETA_SPEC * a.n + i0] = acc;
    yv = yn;
    i0 = i1; i1 = i2;
  }
}

__global__ void theta_lane_major_kernel(const int8_t *__restrict__ yT, int n, int ML, int chunk, int8_t *__restrict__ yTl) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (pos >= ML || i >= n) return;
  const int perc = chunk / 64, ch = pos / chunk, wi = pos - ch * chunk, lane = wi / perc, t = wi - lane * perc;
  const int src = ch * chunk + lane + 64 * t;
  yTl[(int64_t)ML * i + pos] = src < ML ? yT[(int64_t)ML * i + src] : (int8_t)0;
}

int launch_theta_lane_major(hipStream_t st, const int8_t *yT, int n, int ML, int chunk, int8_t *yTl) {
  if (chunk != 512 && chunk != 1024) return GPIRT_ERR_BAD_ARG;
  hipLaunchKernelGGL(theta_lane_major_kernel, dim3((ML + 255) / 256, n), dim3(256), 0, st, yT, n, ML, chunk, yTl);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

__global__ void theta_sum_chunks_kernel(ThetaArgs a, int count) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= count) return;
  const int64_t V = (int64_t)THETA_SPEC * a.n;
  double s = 0.0;
  if (a.gidx[v] >= 0)
    for (int ch = 0; ch < a.nchunks; ++ch) s += a.part[V * (a.slot0 + ch) + v];
  a.buf[v] = s;
}

// items per chunk: C rows of `chunk` doubles in at most 48 KB of LDS
int theta_bucket_chunk(int C) { return min(BUCKET_CHUNK_MAX, max(64, (6144 / C) & ~63)); }

int launch_theta_eval_bucketed(hipStream_t st, const ThetaArgs &a, int initial) {
  if (a.H != 1 || a.G > THETA_BUCKET_MAX_G || !a.tab || !a.part) return GPIRT_ERR_BAD_ARG;
  const int count = initial ? a.n : THETA_SPEC * a.n;
  hipLaunchKernelGGL(theta_bucket_kernel, dim3(1), dim3(1024), 0, st, a, count);
  GP_HIP(hipGetLastError());
  hipLaunchKernelGGL(theta_scatter_kernel, dim3((count + 1023) / 1024), dim3(1024), 0, st, a, count);
  GP_HIP(hipGetLastError());
  const size_t lane_lds = sizeof(double) * (size_t)(a.C + 1) * a.chunk;
  if (a.yTl && a.chunk == 1024 && a.ML % 1024 == 0)
    hipLaunchKernelGGL(theta_eval_bucket_lane_kernel<16>, dim3(a.G, a.nchunks), dim3(256), lane_lds, st, a);
  else if (a.yTl && a.chunk == 512 && a.ML % 512 == 0)
    hipLaunchKernelGGL(theta_eval_bucket_lane_kernel<8>, dim3(a.G, a.nchunks), dim3(256), lane_lds, st, a);
  else
    hipLaunchKernelGGL(theta_eval_bucket_kernel, dim3(a.G, a.nchunks), dim3(256),
                       sizeof(double) * (size_t)a.C * a.chunk, st, a);
  GP_HIP(hipGetLastError());
  if (!a.from_parts) {     // someone other than the decision kernel needs the sums (the exchange)
    hipLaunchKernelGGL(theta_sum_chunks_kernel, dim3((count + 255) / 256), dim3(256), 0, st, a, count);
    GP_HIP(hipGetLastError());
  }
  return GPIRT_OK;
}

int launch_theta_eval(hipStream_t st, const ThetaArgs &a, int initial, bool wide) {
  const int count = initial ? a.n : THETA_SPEC * a.n;
  const dim3 grid(wide ? count : (count + 3) / 4);
  if (a.tab) {
    if (wide) hipLaunchKernelGGL(theta_eval_table_kernel<4>, grid, dim3(256), 0, st, a, initial);
    else hipLaunchKernelGGL(theta_eval_table_kernel<1>, grid, dim3(256), 0, st, a, initial);
  } else {
    if (wide) hipLaunchKernelGGL(theta_eval_kernel<4>, grid, dim3(256), 0, st, a, initial);
    else hipLaunchKernelGGL(theta_eval_kernel<1>, grid, dim3(256), 0, st, a, initial);
  }
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// accept / shrink (elliptical_slice.m:66-100).  q0 = 0 sets the slice height from the
// log-likelihood of the current state; q0 > 0 goes through the round's proposals
// q0, q0+1, ... in order and stops at the first accepted one.  A respondent that is still
// active then gets its next THETA_SPEC proposals, each drawn from the bracket that the
// rejection of the one before would leave (:92-100).
// Four lanes to a respondent, lane k for proposal k: the log-likelihoods are read, the uniforms drawn and the new
// states formed (a sine and a cosine each) side by side; what is sequential -- which proposal is the first to stop
// the chain, how the bracket shrinks from one proposal to the next -- is a ballot and a walk over four shuffled
// numbers that every lane of the four makes for itself.  The arithmetic of a proposal is the one-thread form's,
// operation for operation.  (One thread per respondent: 15.8 us a round, the chain of four Philox blocks, four
// sincos and the dependent loads in between.)
__global__ void theta_decide_kernel(ThetaArgs a, int q0, int round) {
  static_assert(THETA_SPEC == 4, "four lanes to a respondent");
  const int t = blockIdx.x * blockDim.x + threadIdx.x, i = t >> 2, k = t & 3;
  const int qb = (threadIdx.x & 63) & ~3;                   // first lane of this respondent's four
  const int H = a.H, n = a.n;
  const int64_t V = (int64_t)THETA_SPEC * n;
  const bool live = i < n;                                  // the same for the four lanes
  unsigned long long seen = 0;
  auto value = [&](int kk) -> double {                      // the sum a decision reads
    const int64_t v = (int64_t)kk * n + i;
    double ll = 0.0;
    if (!a.from_parts) ll = a.buf[v];
    else                     // bucketed evaluation: the chunk partials, summed in fixed order
      for (int ch = 0; ch < a.nslots; ++ch) ll += a.part[V * ch + v];
    return ll;
  };
  // every sum a decision reads, by its bits and its place (a.ll_seen only)
  auto note = [&](int kk, double ll) {
    const int64_t v = (int64_t)kk * n + i;
    seen += (unsigned long long)__double_as_longlong(ll) * (unsigned long long)(2 * v + 1);
  };
  double lo = 0.0, hi = 0.0, first_phi = 0.0;
  int q = 0;                                                // number of the next proposal
  bool gen = false;                                         // this respondent gets new proposals
  if (q0 == 0) {
    if (live) {
      const double ll = value(0);
      if (k == 0) {
        note(0, ll);
        a.hh[i] = log(rng_uniform_at(a.seed, TAG_T_U, a.sweep, (uint32_t)i, 0)) + ll;
      }
      first_phi = rng_uniform_at(a.seed, TAG_T_U, a.sweep, (uint32_t)i, 1) * kTwoPi;
      lo = first_phi - kTwoPi;
      hi = first_phi;
      q = 1;
      gen = true;
    }
  } else {
    const int nv = live ? a.done[i] : 0;
    const int qk = q0 + k;
    double ll = 0.0, phik = 0.0;
    bool acc = false;
    if (k < nv) {
      ll = value(k);
      acc = ll > a.hh[i];
      phik = a.phi[(int64_t)k * n + i];
    }
    // proposal k ends the chain if it is accepted, if its angle is exactly zero, or if it was the last one allowed
    const bool stop = k < nv && (acc || phik == 0.0 || qk >= a.R);
    const unsigned four = (unsigned)(__ballot(stop) >> qb) & 0xFu;
    const int first = four ? __ffs(four) - 1 : THETA_SPEC;
    if (k < nv && k <= first) note(k, ll);                  // the sums read up to the one that stopped the chain
    const double p0 = __shfl(phik, qb, 64), p1 = __shfl(phik, qb + 1, 64), p2 = __shfl(phik, qb + 2, 64),
                 p3 = __shfl(phik, qb + 3, 64);
    if (first < THETA_SPEC) {
      if (k == first) {
        if (acc)
          for (int h = 0; h < H; ++h)
            a.theta[i + (int64_t)n * h] = a.prop[(int64_t)k * n + i + V * h] + prior_mean(a.tpm, i, h);
        a.done[i] = 0;
        atomicMax(&a.remaining[a.R + 3], qk);
        if (!acc) atomicAdd(&a.counters[1], 1ull);
      }
      if (a.part) a.gidx[(int64_t)k * n + i] = -1;
    } else if (nv > 0) {
      // all rejected: the bracket after their angles, in order
      lo = a.pmin[i];
      hi = a.pmax[i];
      const double ps[THETA_SPEC] = {p0, p1, p2, p3};
#pragma unroll
      for (int kk = 0; kk < THETA_SPEC; ++kk)
        if (kk < nv) {
          if (ps[kk] > 0.0) hi = ps[kk];
          else if (ps[kk] < 0.0) lo = ps[kk];
        }
      q = q0 + nv;           // nv = THETA_SPEC here (a shorter chain ends above)
      gen = true;
    }
  }
  if (gen) {
    if (k == 0) { a.pmin[i] = lo; a.pmax[i] = hi; }
    const int qk = q + k;
    const double u = qk == 1 ? 0.0 : rng_uniform_at(a.seed, TAG_T_U, a.sweep, (uint32_t)i, (uint32_t)qk);
    const double us[THETA_SPEC] = {__shfl(u, qb, 64), __shfl(u, qb + 1, 64), __shfl(u, qb + 2, 64), __shfl(u, qb + 3, 64)};
    int cnt = 0;
    double phi = 0.0, mine = 0.0;
#pragma unroll
    for (int kk = 0; kk < THETA_SPEC; ++kk) {
      if (kk > 0) {            // as if proposal q + kk - 1 had been rejected
        if (phi > 0.0) hi = phi;
        else if (phi < 0.0) lo = phi;
        else break;
        if (q + kk - 1 >= a.R) break;
      }
      phi = q + kk == 1 ? first_phi : us[kk] * (hi - lo) + lo;
      if (kk == k) mine = phi;
      ++cnt;
    }
    const int64_t v = (int64_t)k * n + i;
    if (k < cnt) {
      a.phi[v] = mine;
      double sn, cs;
      sincos(mine, &sn, &cs);
      for (int h = 0; h < H; ++h) a.prop[v + V * h] = a.xc[i + (int64_t)n * h] * cs + a.nu[i + (int64_t)n * h] * sn;
      if (a.part) bucket_note(a, (int)v, a.prop[v] + prior_mean(a.tpm, i, 0));
    } else if (a.part) {
      a.gidx[v] = -1;
    }
    if (k == 0) {
      a.done[i] = cnt;
      atomicAdd(&a.remaining[round], 1);
    }
  }
  if (a.ll_seen) {           // one wrapping sum per wave, then one atomic
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) seen += __shfl_xor(seen, off, 64);
    if ((threadIdx.x & 63) == 0 && seen) atomicAdd(a.ll_seen, seen);
  }
}

int launch_theta_decide(hipStream_t st, const ThetaArgs &a, int q0, int round) {
  hipLaunchKernelGGL(theta_decide_kernel, dim3((THETA_SPEC * a.n + 127) / 128), dim3(128), 0, st, a, q0, round);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// Static respondent step: inverse-CDF draw on the grid (utils/sample_score.m:4-19), H = 1.
//   lp[g,i] = sum_j log P(y_ij | f*_j(x_g))                 (sample_score.m:6-8)
// has only G*m*C distinct cells, so they are tabulated once per sweep (T) and the sum over a
// respondent's items becomes T times the one-hot coding of y -- an FP64 GEMM on the matrix
// cores, G x (C*ML) x n.  The one-hot matrix never changes and is built at set-up.
// ---------------------------------------------------------------------------------------
__global__ void grid_onehot_kernel(const int8_t *__restrict__ yT, int n, int mloc, int ML, int C,
                                   double *__restrict__ oh) {
  const int jl = blockIdx.x * blockDim.x + threadIdx.x, i = blockIdx.y;
  if (jl >= mloc || i >= n) return;
  const int v = yT[(int64_t)ML * i + jl];
  if (v > 0 && v <= C) oh[jl + (int64_t)ML * (v - 1) + (int64_t)ML * C * i] = 1.0;
}

int launch_grid_onehot(hipStream_t st, const int8_t *yT, int n, int mloc, int ML, int C, double *oh) {
  hipLaunchKernelGGL(grid_onehot_kernel, dim3((mloc + 255) / 256, n), dim3(256), 0, st, yT, n, mloc, ML,
                     C, oh);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// T[g + G_pad*(jl + ML*c)] = log-probability of category c+1 for item jl at grid point g
__global__ __launch_bounds__(256) void grid_table_kernel(ThetaArgs a, int G_pad, double *__restrict__ T) {
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  phi_table_to_lds(sphi);
  __syncthreads();
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= a.G) return;
  const int U_pad = a.U_pad;
  const double xg = a.grid_lo + a.grid_step * g;
  // a workgroup stages the tail table once and serves every gridDim.y-th item with it
  for (int jl = blockIdx.y; jl < a.mloc; jl += gridDim.y) {
    const double gg = a.Gt[jl + (int64_t)U_pad * g] + a.betaS[jl] + a.betaS[jl + U_pad] * xg +
                      a.betaS[jl + 2 * (int64_t)U_pad] * xg * xg;
    double lo = a.bthrS[jl];
    for (int c = 0; c < a.C; ++c) {
      const double hi = a.bthrS[jl + (int64_t)U_pad * (c + 1)];
      T[g + (int64_t)G_pad * (jl + (int64_t)a.ML * c)] = ordinal_cell_fast(lo - gg, hi - gg, sphi);
      lo = hi;
    }
  }
}

int launch_grid_table(hipStream_t st, const ThetaArgs &a, int G_pad, double *T) {
  const int nbx = (a.G + 255) / 256, gy = std::max(1, std::min(a.mloc, 2048 / nbx));
  hipLaunchKernelGGL(grid_table_kernel, dim3(nbx, gy), dim3(256), 0, st, a, G_pad, T);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// one wave per respondent: add the prior (sample_score.m:7), cumulate exp(lp) (:10-11), take
// the first grid point whose normalised cumulative sum exceeds the uniform draw (:13-18)
__global__ __launch_bounds__(256) void grid_pick_kernel(ThetaArgs a, const double *__restrict__ lp,
                                                        int G_pad) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int i = blockIdx.x * 4 + w;
  if (i >= a.n) return;
  const int G = a.G, chunk = (G + 63) / 64, g0 = lane * chunk, g1 = min(G, g0 + chunk);
  const double *l = lp + (int64_t)G_pad * i;
  const double s0 = a.tps[2 * i], s1 = a.tps[2 * i + 1];
  const double pv = a.os2 + s0 * s0 + s1 * s1, pm = a.tpm[2 * i] + a.tpm[2 * i + 1];
  auto val = [&](int g) {
    const double xg = a.grid_lo + a.grid_step * g - pm;
    return l[g] + -0.5 * xg * xg / pv;
  };
  double mx = -INFINITY;
  for (int g = g0; g < g1; ++g) mx = fmax(mx, val(g));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmax(mx, __shfl_xor(mx, off, 64));
  double mine = 0.0;
  for (int g = g0; g < g1; ++g) mine += exp(val(g) - mx);
  double incl = mine;                       // inclusive scan over the lanes
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const double t = __shfl_up(incl, off, 64);
    if (lane >= off) incl += t;
  }
  const double total = __shfl(incl, 63, 64);
  const double c0 = exp(val(0) - mx), span = total - c0;
  const double r = rng_uniform_at(a.seed, TAG_T_U, a.sweep, (uint32_t)i, 0);
  double c = incl - mine;
  int first = G - 1;
  for (int g = g0; g < g1; ++g) {
    c += exp(val(g) - mx);
    if ((c - c0) / span > r) { first = g; break; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
  if (lane == 0) a.theta[i] = a.grid_lo + a.grid_step * first;
}

int launch_grid_pick(hipStream_t st, const ThetaArgs &a, const double *lp, int G_pad) {
  hipLaunchKernelGGL(grid_pick_kernel, dim3((a.n + 3) / 4), dim3(256), 0, st, a, lp, G_pad);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// Running sums for the posterior summaries (gpirt-sythetic.R:111-133,151-214; the table
// behind getprobs_gpirt, getprob_gpirt.R:1-20)
// ---------------------------------------------------------------------------------------
__global__ void summary_theta_kernel(const double *__restrict__ theta, int64_t count,
                                     double *__restrict__ sum, double *__restrict__ sq) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const double t = theta[i];
  sum[i] += t;
  sq[i] += t * t;
}

int launch_summary_theta(hipStream_t st, const double *theta, int64_t count, double *sum, double *sq) {
  hipLaunchKernelGGL(summary_theta_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, theta,
                     count, sum, sq);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

__global__ __launch_bounds__(256) void summary_prob_kernel(ThetaArgs a, int quadratic,
                                                           double *__restrict__ psum,
                                                           double *__restrict__ asum) {
  __shared__ __attribute__((aligned(16))) double sphi[PHI_LDS_DOUBLES];
  phi_table_to_lds(sphi);
  __syncthreads();
  const int u = blockIdx.x * blockDim.x + threadIdx.x;
  const int U_pad = a.U_pad;
  if (u >= U_pad || u % a.ML >= a.mloc) return;
  // a workgroup stages the tail table once and serves every gridDim.y-th grid point with it
  for (int g = blockIdx.y; g < a.G; g += gridDim.y) {
    const double xg = a.grid_lo + a.grid_step * g;
    double F = a.Gt[u + (int64_t)U_pad * g] + a.betaS[u] + a.betaS[u + U_pad] * xg;
    if (quadratic) F += a.betaS[u + 2 * (int64_t)U_pad] * xg * xg;
    double lo = a.bthrS[u], best = -1.0;
    int arg = 1;
    for (int c = 0; c < a.C; ++c) {
      const double hi = a.bthrS[u + (int64_t)U_pad * (c + 1)];
      const double p = ordinal_cell_prob(lo - F, hi - F, sphi);
      psum[u + (int64_t)U_pad * (c + (int64_t)a.C * g)] += p;
      if (p > best) { best = p; arg = c + 1; }      // which.max: the first maximum
      lo = hi;
    }
    asum[u + (int64_t)U_pad * g] += (double)arg;
  }
}

int launch_summary_prob(hipStream_t st, const ThetaArgs &a, int quadratic, double *psum, double *asum) {
  const int nbx = (a.U_pad + 255) / 256, gy = std::max(1, std::min(a.G, 2048 / nbx));
  hipLaunchKernelGGL(summary_prob_kernel, dim3(nbx, gy), dim3(256), 0, st, a,
                     quadratic, psum, asum);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

// ---------------------------------------------------------------------------------------
// snap: re-anchor f at the new theta from f* (main.m:69-73).  Gt is unit-contiguous, f is
// row-contiguous: a 64x64 tile goes through LDS so that both sides are coalesced.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void snap_kernel(const double *__restrict__ Gt, int U_pad,
                                                   const double *__restrict__ x, int64_t x_stride,
                                                   int Np, int mloc, int ML, int G, double lo,
                                                   double step, int truncate, double *__restrict__ f,
                                                   int64_t ld) {
  __shared__ double tile[64][65];
  __shared__ int gidx[64];
  const int p = blockIdx.z;
  const int r0 = blockIdx.x * 64, u0 = blockIdx.y * 64;  // u0 within the group's mloc units
  const int tid = threadIdx.x;
  if (tid < 64) {
    int r = r0 + tid;
    gidx[tid] = r < Np ? grid_index(x[x_stride * p + r], lo, step, G, truncate) : 0;
  }
  __syncthreads();
  const int ul = tid & 63;
  for (int rl = tid >> 6; rl < 64; rl += 4) {
    int uu = u0 + ul;
    tile[rl][ul] = uu < mloc ? Gt[(int64_t)U_pad * gidx[rl] + (int64_t)ML * p + uu] : 0.0;
  }
  __syncthreads();
  const int rl2 = tid & 63;
  for (int ul2 = tid >> 6; ul2 < 64; ul2 += 4) {
    int r = r0 + rl2, uu = u0 + ul2;
    if (r < Np && uu < mloc) f[r + ld * ((int64_t)ML * p + uu)] = tile[rl2][ul2];
  }
}

int launch_snap(hipStream_t st, const double *Gt, int U_pad, const double *x, int64_t x_stride,
                int Np, int P, int mloc, int ML, int G, double lo, double step, int truncate,
                double *f, int64_t ld) {
  dim3 grid((Np + 63) / 64, (mloc + 63) / 64, P);
  hipLaunchKernelGGL(snap_kernel, grid, dim3(256), 0, st, Gt, U_pad, x, x_stride, Np, mloc, ML, G, lo,
                     step, truncate, f, ld);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

__global__ __launch_bounds__(256) void export_fstar_kernel(const double *__restrict__ Gt, int U_pad,
                                                           int G, int mloc, int ML,
                                                           double *__restrict__ out) {
  __shared__ double tile[64][65];
  const int p = blockIdx.z;
  const int g0 = blockIdx.x * 64, u0 = blockIdx.y * 64, tid = threadIdx.x;
  const int ul = tid & 63;
  for (int gl = tid >> 6; gl < 64; gl += 4) {
    int g = g0 + gl, u = u0 + ul;
    tile[gl][ul] = (g < G && u < mloc) ? Gt[u + (int64_t)ML * p + (int64_t)U_pad * g] : 0.0;
  }
  __syncthreads();
  const int gl2 = tid & 63;
  for (int ul2 = tid >> 6; ul2 < 64; ul2 += 4) {
    int g = g0 + gl2, u = u0 + ul2;
    if (g < G && u < mloc) out[g + (int64_t)G * (u + (int64_t)mloc * p)] = tile[gl2][ul2];
  }
}

int launch_export_fstar(hipStream_t st, const double *Gt, int U_pad, int G, int P, int mloc, int ML,
                        double *out) {
  dim3 grid((G + 63) / 64, (mloc + 63) / 64, P);
  hipLaunchKernelGGL(export_fstar_kernel, grid, dim3(256), 0, st, Gt, U_pad, G, mloc, ML, out);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

}  // namespace gpirt
