// Argument blocks and launch prototypes of the per-unit and per-respondent kernels.
#pragma once
#include "common.h"

namespace gpirt {

constexpr int MAXC = 16;   // response categories supported on the device
constexpr int MAXH = 64;   // periods supported by the respondent step

// ---- ess.hip --------------------------------------------------------------------------
int launch_loglik(hipStream_t st, const double *g, int64_t ldg, const int8_t *y, int64_t ldy,
                  const double *bthr, int n, int U, int C, double *ll_unit, double *ll_row);

int launch_fill_normal(hipStream_t st, double *Z, int64_t ld, int Np, int Np_pad, int U_pad,
                       int mloc, int ML, int m, int j0, uint64_t seed, uint32_t sweep);

int launch_rng_fill(hipStream_t st, uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                    int64_t count, int normal, double *out);

struct EssFArgs {
  double *f; const double *nu; int64_t ld;   // [ld, U] latent state and ellipse draw
  const double *mu; int64_t ld_mu;           // explicit mean (test entry) or nullptr
  const int8_t *y; int64_t ldy;              // [ldy, U], 0 = missing
  const double *x; int64_t x_stride;         // inputs of group p at x + x_stride*p (or nullptr)
  const double *beta;                        // [3,U] (or nullptr with mu)
  const double *bthr;                        // [C+1,U], ends ignored
  const double *unif;                        // injected uniforms [R+2,U] or nullptr (Philox)
  int Np, U_pad, C, mloc, ML, m, j0, R;   // unit u = jl + ML*p is live when jl < mloc
  uint64_t seed; uint32_t sweep;
  int32_t *rounds; double *ll;               // per-unit outputs (optional)
  unsigned long long *counters;              // [0] likelihood evaluations, [1] capped updates
  int wide8;                                 // eight waves per unit instead of four (few units of many rows)
  int u0, ucount;                            // a launch of the units u0 .. u0 + ucount - 1 (ucount = 0: all)
};
int launch_ess_f(hipStream_t st, const EssFArgs &a);

struct EssParamArgs {
  const double *f; int64_t ld;
  const int8_t *y; int64_t ldy;
  const double *x; int64_t x_stride;
  double *ebeta;        // [3,U]   whitened coefficients (state)
  double *delta;        // [C-1,U] unconstrained thresholds (state)
  double *beta;         // [3,U]   out
  double *bthr;         // [C+1,U] out (+-inf ends)
  double *betaS;        // [U_pad,3]   out, unit-contiguous copy for the respondent step
  double *bthrS;        // [U_pad,C+1] out, unit-contiguous copy
  const double *bpm, *bps;  // [3,mloc] prior mean / sd
  int fixed_thr;
  int Np, U_pad, C, mloc, ML, m, j0, R;
  uint64_t seed; uint32_t sweep;
  double *ll;
  const double *ll_in;   // log-likelihood of the current state if the caller has it (else null)
  unsigned long long *counters;
  int u0, ucount;                            // as in EssFArgs
};
int launch_ess_param(hipStream_t st, const EssParamArgs &a);

// ---- theta.hip -------------------------------------------------------------------------
struct ThetaArgs {
  double *theta;            // [n,H] state
  double *xc, *nu;          // [n,H] centred state, ellipse draw
  double *prop;             // [SPEC*n, H] centred proposals: entry k*n + i is respondent i's k-th of the round
  double *hh, *pmin, *pmax; // [n] slice height, bracket before the round's proposals
  double *phi;              // [SPEC*n] angles of the round's proposals
  int32_t *done;            // [n] proposals of respondent i in flight this round (0: finished)
  double *buf;              // [SPEC*n] log-likelihood per proposal (partial, then all-reduced)
  const double *Lt;         // [H,H] lower factor of the temporal kernel
  const double *tpm, *tps;  // [2,n]
  const double *Gt;         // [U_pad, G_pad] f* with the unit index contiguous
  const double *betaS, *bthrS;
  const int8_t *yT;         // [ML*H, n] responses, respondent-major
  const int8_t *yTl;        // the same with the items of a chunk lane-major (theta_lane_major), or nullptr
  double *tab;              // [U_pad, C, G] log cell probabilities on the grid (table variant) or null
  // bucketed evaluation of the table variant (H = 1): respondents grouped by grid index
  int32_t *gidx, *order;    // [SPEC*n] grid index of the evaluated theta (-1: none), entries by bucket
  int32_t *bstart;          // [G+1] bucket g holds order[bstart[g] .. bstart[g+1])
  int32_t *bcur;            // [G] next free place of bucket g while the entries are being filed
  int32_t *ghist;           // [G] respondents per grid index, counted by the prepare / decision kernels
  double *part;             // [nslots, SPEC*n] partial sums per item chunk
  int chunk, nchunks;       // items per chunk (multiple of 64), number of this rank's chunks
  int slot0, nslots;        // `part` holds nslots chunks; this rank's are slot0 .. slot0 + nchunks - 1 (the
                            // others are those of the other ranks when block sums are exchanged)
  int from_parts;           // the decision kernel sums `part` itself instead of reading `buf`
  int n, H, C, mloc, ML, U_pad, G, constant_irf, R;
  double grid_lo, grid_step;
  int index_truncate;       // grid index = floor instead of nearest
  double os2;               // prior variance of the static step (theta_os^2, 1 when theta_os = 0)
  uint64_t seed; uint32_t sweep;
  int32_t *remaining;       // [R+4] respondents still active after the decision of round r; the last
                            // entry: largest proposal number at which a respondent finished
  unsigned long long *counters;  // [1] capped
  unsigned long long *ll_seen;   // optional: wrapping sum over every log-likelihood sum a decision read of
                                 // (its bit pattern) x (2 entry + 1): equal chains of decisions on equal bits
                                 // give equal words whatever the order of the atomics
};
// Proposals per lock-step round.  After a rejection the next angle is a function of the
// bracket and the generator alone, so a round draws each respondent's next THETA_SPEC
// proposals as if all were rejected, evaluates them together and then takes the first one
// that is accepted: same draws and decisions as one proposal per round, a quarter of the
// rounds (launch gaps, host looks and, with sharded items, all-reduce latencies).
constexpr int THETA_SPEC = 4;
int launch_theta_prepare(hipStream_t st, const ThetaArgs &a);
int launch_theta_eval(hipStream_t st, const ThetaArgs &a, int initial, bool wide);
// tab[u + U_pad*(c + C*g)] = log P(y = c+1 | F*_u(x_g)); with a.tab set, launch_theta_eval sums
// table entries instead of evaluating cells
int launch_theta_table(hipStream_t st, const ThetaArgs &a);
// H = 1, table variant: group the active respondents by the grid index of the theta to be
// evaluated; every (grid index, item chunk) stages its table rows in LDS once and serves all
// respondents of the bucket; buf[i] = sum of the chunk partials in fixed order
constexpr int THETA_BUCKET_MAX_G = 2048;
int theta_bucket_chunk(int C);
int launch_theta_eval_bucketed(hipStream_t st, const ThetaArgs &a, int initial);
// yTl[i][ch*chunk + lane*(chunk/64) + t] = yT[i][ch*chunk + lane + 64 t]; chunk must be 512 or 1024
int launch_theta_lane_major(hipStream_t st, const int8_t *yT, int n, int ML, int chunk, int8_t *yTl);
// q0 = 0: set the slice height from the current state's log-likelihood and draw proposals
// 1..SPEC; q0 > 0: the round evaluated proposals q0..q0+SPEC-1.  round = index into remaining.
int launch_theta_decide(hipStream_t st, const ThetaArgs &a, int q0, int round);
// static (grid) respondent step, H = 1: one-hot coding of y (k = jl + ML*(y-1), ld = C*ML),
// the per-sweep table of log cell probabilities T[G_pad, C*ML], and the inverse-CDF draw
// from lp = T * onehot (+ prior)
int launch_grid_onehot(hipStream_t st, const int8_t *yT, int n, int mloc, int ML, int C, double *oh);
int launch_grid_table(hipStream_t st, const ThetaArgs &a, int G_pad, double *T);
int launch_grid_pick(hipStream_t st, const ThetaArgs &a, const double *lp, int G_pad);
// running sums for the posterior summaries: theta and theta^2; per (unit, grid point) the
// category probabilities psum[u + U_pad*(c + C*g)] and the most probable category
// asum[u + U_pad*g], from F* = Gt + beta0 + beta1 x (+ beta2 x^2)
int launch_summary_theta(hipStream_t st, const double *theta, int64_t count, double *sum, double *sq);
int launch_summary_prob(hipStream_t st, const ThetaArgs &a, int quadratic, double *psum, double *asum);
// f[r + ld*u] = Gt[u + U_pad * idx(x_p[r])]   (main.m:69-73)
int launch_snap(hipStream_t st, const double *Gt, int U_pad, const double *x, int64_t x_stride,
                int Np, int P, int mloc, int ML, int G, double lo, double step, int truncate,
                double *f, int64_t ld);
// out[g + G*(jl + mloc*p)] = Gt[(jl + ML*p) + U_pad*g]
int launch_export_fstar(hipStream_t st, const double *Gt, int U_pad, int G, int P, int mloc, int ML,
                        double *out);

}  // namespace gpirt
