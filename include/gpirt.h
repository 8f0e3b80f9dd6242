/*
 * gpirt.h -- plain C ABI of the MI355X-native GPIRT Gibbs sweep.
 *
 * This is the boundary an R `.Call` shim (or any other FFI: ctypes, cgo, JNI)
 * binds.  It replaces the R-level call
 *     gpirtMCMC(data, sample_iterations, burn_iterations, THIN, CHAIN, ...)
 * made by the reference drivers (gpirt-sythetic.R:88-94, gpirt-senate.R:77-81,
 * TAPS/gpirt_TAPS.R:30-36, VAA/gpirt_VAA.R:137-145), whose sampler lives in an
 * external Rcpp package; the in-tree statement of the same sweep is the MATLAB
 * prototype main.m:46-87 + utils/{elliptical_slice,sample_fs,log_like_fn,
 * sigmoid_fn,sample_score,model}.m.
 *
 * Conventions
 *   - No C++ types, no torch types, no exceptions cross this boundary.
 *   - Every array is column-major (R order).  `data[n,m,H]` element (i,j,h) is
 *     at i + n*(j + m*h).
 *   - The caller owns every buffer it passes.  The library never frees caller
 *     memory and returns no library-owned pointer except static strings.
 *   - Every function returns GPIRT_OK (0) or a negative gpirt_status; the text
 *     of the last failure on the calling thread is gpirt_last_error().
 *   - Pointers named `h_*`/unprefixed are HOST pointers; `d_*` are DEVICE
 *     pointers (hipMalloc memory on the current device).
 */
#ifndef GPIRT_H
#define GPIRT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPIRT_VERSION_MAJOR 0
#define GPIRT_VERSION_MINOR 1

typedef enum gpirt_status {
  GPIRT_OK = 0,
  GPIRT_ERR_BAD_ARG = -1,     /* bad shape / null pointer / unsupported option   */
  GPIRT_ERR_NOT_PD = -2,      /* Cholesky failed even after jitter escalation    */
  GPIRT_ERR_ESS_COLLAPSE = -3,/* slice bracket shrank to the current point
                                 (utils/elliptical_slice.m:96-98 raises here)    */
  GPIRT_ERR_HIP = -4,         /* a HIP runtime call failed                       */
  GPIRT_ERR_COMM = -5,        /* the collective (RCCL or callback) failed        */
  GPIRT_ERR_NO_DEVICE = -6,   /* no gfx950 device visible                        */
  GPIRT_ERR_INTERRUPT = -7    /* the progress callback asked to stop             */
} gpirt_status;

/* KERNEL argument of gpirtMCMC ("RBF" gpirt-sythetic.R:93, "Matern" TAPS/onerun:23):
 * temporal kernel of the respondent trajectories. */
typedef enum gpirt_kernel { GPIRT_KERNEL_RBF = 0, GPIRT_KERNEL_MATERN52 = 1 } gpirt_kernel;

/* How the respondent step draws theta. */
typedef enum gpirt_theta_mode {
  GPIRT_THETA_ESS = 0,  /* elliptical slice over each respondent's H-vector under the
                           temporal GP prior, likelihood by grid lookup into f*       */
  GPIRT_THETA_GRID = 1  /* inverse-CDF draw on the grid (utils/sample_score.m:4-19);
                           H = 1 only                                                 */
} gpirt_theta_mode;

enum {
  GPIRT_RETURN_F = 1,      /* copy each kept draw of f back (large: n*m*H doubles per draw) */
  GPIRT_RETURN_FSTAR = 2,  /* copy each kept draw of f* back (n_grid*m*P doubles per draw)  */
  GPIRT_EXACT_FSTAR = 8,   /* evaluate K(x, xs) at every grid point in the f* step instead of
                              interpolating it from Chebyshev nodes (see gpirt_sampler_sweep)  */
  GPIRT_ALLREDUCE_ON_STREAM = 4, /* the all-reduce hook ENQUEUES the collective on the sampler's
                              HIP stream (gpirt_sampler_stream) and returns without waiting;
                              the library then does not drain the stream around the call and
                              looks at the round counters only every few rounds               */
  GPIRT_THETA_TABLE = 16,  /* respondent slice step: tabulate the log cell probabilities on the
                              grid once per sweep and sum table entries (default from 1024
                              respondents on; the flag forces it)                             */
  GPIRT_THETA_DIRECT = 32, /* respondent slice step: evaluate every cell in every round (default
                              below 1024 respondents; the flag forces it)                     */
  GPIRT_FSTAR_SOLVE = 64,  /* f* step: keep the interpolated cross-covariance but go through the
                              two triangular solves with the n x n factor instead of the low-rank
                              identity (see gpirt_sampler_sweep)                              */
  GPIRT_NO_NU_OVERLAP = 256, /* compute nu = L Z in one launch after the factorisation; by default,
                              from 4096 inputs on, the first rows of it are computed beside the
                              factorisation on a second stream (same result bit for bit) when the
                              sampler's own timing of its first sweeps finds that faster (it
                              times up to four schedules, three sweeps each; the flag takes the
                              two with rows of nu beside the factorisation out of the choice)   */
  GPIRT_GENERATOR_PRIOR = 128, /* keep the Cholesky factor of K(theta,theta) + s2 I in generator
                              (block semiseparable) form instead of as an n x n matrix (see
                              gpirt_sampler_sweep); one IRF group only, ignored otherwise      */
  GPIRT_INDEX_TRUNCATE = 512, /* look f* up at the grid point at or below theta, the rule of the
                              R post-processing (gpirt-sythetic.R:190), instead of at the
                              nearest one (main.m:69, the default)                             */
  GPIRT_FIXED_ORDER_SUM = 1024 /* respondent slice step with sharded items: exchange the partial
                              sums of fixed blocks of GPIRT_SUM_BLOCK items (not each rank's
                              total) and add the blocks in item order on every rank, so that the
                              chain is the same BIT FOR BIT for every number of ranks, one
                              included.  Needs H = 1, the tabulated variant (GPIRT_THETA_TABLE or
                              >= 1024 respondents) and item ranges that begin and end on block
                              boundaries (or at the last item); refused otherwise.  The hook then
                              carries ceil(m / GPIRT_SUM_BLOCK) times the doubles per call.      */
};
#define GPIRT_SUM_BLOCK 512

/* Sum-all-reduce hook used by the respondent step when items are sharded over
 * ranks: must leave, in place, the element-wise sum over all ranks of `count`
 * doubles at `buf` on every rank.  `buf` is a DEVICE pointer for the HIP
 * backend and a host pointer for the CPU oracle.  Return 0 on success.  Unless
 * GPIRT_ALLREDUCE_ON_STREAM is set, the sampler's stream is idle during the call and
 * the result must be complete when the hook returns. */
typedef int (*gpirt_allreduce_fn)(double *buf, int64_t count, void *user);

/* Called once per finished sweep on the calling thread (the place for
 * R_CheckUserInterrupt); a non-zero return aborts with GPIRT_ERR_INTERRUPT. */
typedef int (*gpirt_progress_fn)(int32_t sweep, int32_t total_sweeps, void *user);

/* Named arguments of gpirtMCMC (union over the reference's call sites) plus the
 * knobs the reference hard-codes.  Zero-initialise, then call
 * gpirt_config_defaults(). */
typedef struct gpirt_config {
  int32_t struct_size;        /* = sizeof(gpirt_config), ABI check                         */
  int32_t n;                  /* respondents                                               */
  int32_t m;                  /* items (GLOBAL count, also when sharded)                   */
  int32_t H;                  /* periods (horizon)                                         */
  int32_t C;                  /* response categories; data values are 1..C, 0 = missing    */
  int32_t sample_iterations;  /* gpirt-sythetic.R:60                                       */
  int32_t burn_iterations;    /* gpirt-sythetic.R:61                                       */
  int32_t thin;               /* THIN, gpirt-sythetic.R:75                                 */
  int32_t chain;              /* chain index 0..CHAIN-1; independent stream per chain      */
  uint64_t seed;              /* SEED                                                      */
  double theta_os;            /* temporal output scale   (gpirt-sythetic.R:62-73)          */
  double theta_ls;            /* temporal length scale; <= 0 => constant over time
                                 (gpirt-senate.R:22-28)                                    */
  int32_t kernel;             /* gpirt_kernel                                              */
  int32_t constant_irf;       /* 1: one IRF per item shared by all periods
                                 (TAPS/gpirt_TAPS.R:36)                                    */
  double jitter;              /* added to diag K(theta,theta); 1e-6 (generate_data.R:55)   */
  double irf_ls;              /* IRF kernel length scale, 1 (utils/model.m:6)              */
  double irf_sf;              /* IRF kernel signal sd, 1     (utils/model.m:6)             */
  int32_t n_grid;             /* 1001: xs = seq(-5,5,0.01) (gpirt-sythetic.R:113)          */
  double grid_lo;             /* -5                                                        */
  double grid_step;           /* 0.01                                                      */
  int32_t theta_mode;         /* gpirt_theta_mode                                          */
  int32_t max_ess_rounds;     /* cap on shrink rounds per slice update (0 => 200)          */
  int32_t flags;              /* GPIRT_RETURN_*                                            */
  int32_t item_begin;         /* this rank owns global items [item_begin, item_end)        */
  int32_t item_end;           /* 0,0 => all items                                          */
  int32_t rank;
  int32_t world;              /* number of ranks sharing the chain (1 => no collective)    */
  gpirt_allreduce_fn allreduce; /* required when world > 1                                 */
  void *allreduce_user;
  gpirt_progress_fn progress; /* optional                                                  */
  void *progress_user;
} gpirt_config;

/* Caller-allocated output buffers; any pointer may be NULL to skip that output.
 * K = sample_iterations / thin kept draws, m_loc = item_end - item_begin,
 * P = constant_irf ? 1 : H.  The draw index is the slowest dimension. */
typedef struct gpirt_outputs {
  double *theta;      /* [n, H, K]                 samples$theta (gpirt-sythetic.R:123)        */
  double *f;          /* [n, m_loc, H, K]          samples$f, non-linear residual (:153)       */
  double *fstar;      /* [n_grid, m_loc, P, K]     samples$fstar on the grid (:154)            */
  double *beta;       /* [3, m_loc, P, K]          samples$beta (:153)                         */
  double *threshold;  /* [m_loc, C+1, P, K]        samples$threshold, +-Inf ends (:193)        */
  double *loglik;     /* [sample+burn]             local-items log-likelihood after each sweep */
  /* diagnostics (totals over the whole run) */
  int64_t ess_f_rounds;       /* likelihood evaluations spent in the f step             */
  int64_t ess_param_rounds;   /* ... in the beta/threshold step                         */
  int64_t ess_theta_rounds;   /* ... in the respondent step (lock-step rounds)          */
  int64_t ess_capped;         /* slice updates that hit max_ess_rounds (state kept)     */
  int32_t jitter_retries;     /* Cholesky retries with 10x jitter                       */
  int32_t nu_rows_beside_factor; /* rows of nu = L Z computed beside the factorisation in the
                                    last sweep (0: one launch after it)                    */
  int32_t factor_lookahead;   /* 1: the last sweep's factorisation ran its trailing updates
                                 on the second stream beside the next block columns' steps */
  int32_t nu_tail_panels;     /* 512-column panels of L whose share of the rows of nu AFTER
                                 nu_rows_beside_factor was added beside the factorisation too */
  int32_t slice_second_share; /* units (items x IRF groups) whose slice updates the last sweep ran as a
                                 second share, on a stream of their own beside the first share's
                                 next step (0: one launch per step); GPIRT_SLICE_SHARES sets it */
  double seconds_total;       /* wall time inside the sweep loop                        */
} gpirt_outputs;

/* ---- library -------------------------------------------------------------------- */

/* "gpirt-hip <major>.<minor> (gfx950)" -- static string. */
const char *gpirt_version(void);
/* Text of the last failure on this thread -- static thread-local buffer. */
const char *gpirt_last_error(void);
/* Number of visible HIP devices (0 when there is no GPU or no driver). */
int gpirt_device_count(void);
/* Select the device used by every later call on this thread. */
int gpirt_set_device(int device);
/* Copy `bytes` between host and device memory with the HIP runtime this library runs on
 * (to_device = 1: host -> device; 0: device -> host); blocking.  No counterpart in the
 * reference: for all-reduce hooks that stage the exchange through host memory without a
 * HIP binding of their own (gpirt_amd/dist.py:host_staged_allreduce). */
int gpirt_device_copy(void *dst, const void *src, int64_t bytes, int32_t to_device);
/* Fill `cfg` with the reference's defaults (gpirt-sythetic.R:60-94, utils/model.m:6). */
void gpirt_config_defaults(gpirt_config *cfg);

/* ---- the drop-in: one chain of gpirtMCMC ------------------------------------------
 * Replaces gpirtMCMC(...)[[chain]] (gpirt-sythetic.R:88-94); the sweep follows
 * main.m:46-87 generalised to ordinal probit / thresholds / dynamic theta as the R
 * call sites require.
 *   y                  int8 [n, m, H], 1..C, 0 = NA           (`data`)
 *   theta_init         [n, H] or NULL (=> N(0,1) draws)       (`theta_init`)
 *   beta_prior_means   [3, m] or NULL (=> 0)
 *   beta_prior_sds     [3, m] or NULL (=> 1); 0 switches a term off (TAPS/gpirt_TAPS.R:28-29)
 *   theta_prior_means  [2, n] or NULL (=> 0)   constant / linear-trend mean
 *   theta_prior_sds    [2, n] or NULL (=> 0)   constant / linear-trend sd
 *   thresholds         [C+1] fixed thresholds (first/last ignored, treated as -+Inf),
 *                      or NULL => sampled per (item, period) (gpirt-sythetic.R:94)
 * All prior arrays index GLOBAL items.  y holds all m items; a sharded rank reads
 * only its own columns. */
int gpirt_run(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
              const double *beta_prior_means, const double *beta_prior_sds,
              const double *theta_prior_means, const double *theta_prior_sds,
              const double *thresholds, gpirt_outputs *out);

/* ---- resident sampler (what gpirt_run is built from) ---------------------------------
 * Keeps y, theta, f, L, ... in HBM between calls so that a caller (the benchmark, an
 * interactive R session) can advance the chain without re-uploading anything. */
typedef struct gpirt_sampler gpirt_sampler;

int gpirt_sampler_create(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
                         const double *beta_prior_means, const double *beta_prior_sds,
                         const double *theta_prior_means, const double *theta_prior_sds,
                         const double *thresholds, gpirt_sampler **out);
/* Advance `sweeps` Gibbs sweeps (main.m:46-87 body).  Asynchronous work is drained
 * before return.
 *
 * The f* step (main.m:57-58, f* = K(xs, x) [K(x, x) + s2 I]^-1 f) has three routes that agree
 * to the accuracy the condition of K(x, x) + s2 I allows (about 1e-6 of |f*| at n = 10 000):
 *   default            the kernel is reproduced to rounding level from 128 Chebyshev nodes in
 *                      both arguments, K(x, x') = l(x)^T K(c, c) l(x'), which turns the step
 *                      into one [units x n x 128] product and a 128 x 128 factorisation; used
 *                      while every input lies on the grid's range, else FSTAR_SOLVE's route;
 *   GPIRT_FSTAR_SOLVE  K(x, xs) = K(x, c) Lag interpolated in the grid argument only, two
 *                      triangular solves with the n x n factor for the 128 node columns;
 *   GPIRT_EXACT_FSTAR  K(x, xs) evaluated at every grid point, solves for all columns.
 *
 * The prior draw nu = chol(K(x, x) + s2 I) z (sample_fs.m:2, demo.m:45) has two routes:
 *   default                the n x n matrix is built and factorised (blocked, FP64 MFMA) and
 *                          nu = L Z is one triangular product for all items;
 *   GPIRT_GENERATOR_PRIOR  with the same node representation, K + s2 I = Phi Phi^T + s2 I has
 *                          a factor whose blocks below the diagonal are Phi_i W_b: only the
 *                          128 x 128 diagonal blocks and the generators W_b are computed
 *                          (batched), nu costs O(n 128 m) instead of n^2 m, and the n x n matrix
 *                          is never formed.  Same nu to about 1e-8 (the factor of a matrix that
 *                          equals K + s2 I to 3e-15); a sweep whose inputs leave the grid's
 *                          range takes the default route. */
int gpirt_sampler_sweep(gpirt_sampler *s, int32_t sweeps);
/* Copy the current state to host buffers (any may be NULL):
 * theta [n,H]; f [n,m_loc,H]; fstar [n_grid,m_loc,P]; beta [3,m_loc,P];
 * threshold [m_loc,C+1,P]. */
int gpirt_sampler_get_state(gpirt_sampler *s, double *theta, double *f, double *fstar,
                            double *beta, double *threshold);
/* Per-phase device time accumulated since creation (ms) and launch counts, for the
 * roofline report.  Phases: 0 k_build, 1 potrf, 2 nu=L*Z (one GEMM launch per sweep),
 * 3 ess_f, 4 ess_param, 5 fstar solves+gemm, 6 theta step, 7 snap, 8 normal draws.
 * ms[9], count[9]. */
int gpirt_sampler_phase_times(gpirt_sampler *s, double *ms, int64_t *count);
/* The rank-512 trailing updates of the blocked Cholesky (the FP64 MFMA launches that carry most of
 * its n^3/3 flop; the last few, with at most 300 tiles, run on another kernel and are left out), each
 * bracketed by HIP events on the sampler's stream: device time summed over
 * the launches since creation (ms), their number, and their algorithmic flop (lower tiles only). */
int gpirt_sampler_update_times(gpirt_sampler *s, double *ms, int64_t *launches, double *flop);
/* Diagnostics so far: fills the counters of `out` (buffers untouched). */
int gpirt_sampler_diagnostics(gpirt_sampler *s, gpirt_outputs *out);
/* GPIRT_FIXED_ORDER_SUM only: a 64-bit checksum of every per-respondent log-likelihood sum that
 * the respondent step's decisions have read since creation -- the wrapping sum of (bit pattern of
 * the sum) x (2 x entry + 1) over all of them.  Chains whose sums agree bit for bit, for whatever
 * number of ranks, return the same word; a single differing bit changes it.  No counterpart in the
 * reference (single process, nothing to compare). */
int gpirt_sampler_ll_checksum(gpirt_sampler *s, uint64_t *out);
/* ---- posterior summaries from running sums on the device ---------------------------
 * What the reference's R post-processing derives from the kept draws
 * (gpirt-sythetic.R:111-133,151-214,231-251; getprob_gpirt.R:1-20), accumulated on the
 * device so that the f / f* draws need not leave it:
 *   theta_mean, theta_sd   pred_theta / pred_theta_sd before centring (:122-123), sd with
 *                          the n-1 divisor of R's sd(); over all kept draws, or over all but the
 *                          first as the reference's `[-1, i, h]` does (GPIRT_SUMMARY_DROP_FIRST_THETA)
 *   prob_mean              mean over draws of P(y = c | x_g) =
 *                          Phi(b_c - F*(x_g)) - Phi(b_{c-1} - F*(x_g)) with
 *                          F* = fstar + beta_0 + beta_1 x_g (:153-154; `quadratic` adds
 *                          beta_2 x_g^2, which the reference's post-processing leaves out):
 *                          the table behind getprobs_gpirt's ICC and the predictive
 *                          log-likelihood at any grid index
 *   argmax_mean            mean over draws of which.max_c of that probability (:197,208)
 * Buffers are caller-allocated; any pointer may be NULL. */
typedef struct gpirt_summary {
  int32_t struct_size;  /* = sizeof(gpirt_summary)                                  */
  int32_t draws;        /* out: number of accumulated draws                         */
  double *theta_mean;   /* [n, H]                                                   */
  double *theta_sd;     /* [n, H]                                                   */
  double *prob_mean;    /* [n_grid, m_loc, P, C]                                    */
  double *argmax_mean;  /* [n_grid, m_loc, P]   (categories counted from 1)         */
} gpirt_summary;
/* Options of the two calls below (OR-ed). */
enum {
  GPIRT_SUMMARY_QUADRATIC = 1,        /* F* also gets beta_2 x_g^2                                    */
  GPIRT_SUMMARY_SKIP_THETA = 2,       /* gpirt_sampler_accumulate: leave theta out of this draw's sums */
  GPIRT_SUMMARY_DROP_FIRST_THETA = 4  /* gpirt_run_summary: the theta sums leave out the first kept
                                         draw, as samples$theta[-1, i, h] does (gpirt-sythetic.R:123) */
};
/* Add the sampler's current state to the running sums (call after a kept sweep). */
int gpirt_sampler_accumulate(gpirt_sampler *s, int32_t options);
/* Finish the running sums into `out` (they keep accumulating afterwards). */
int gpirt_sampler_summary(gpirt_sampler *s, gpirt_summary *out);
/* gpirt_run that also accumulates every kept draw; `summary` may be NULL. */
int gpirt_run_summary(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
                      const double *beta_prior_means, const double *beta_prior_sds,
                      const double *theta_prior_means, const double *theta_prior_sds,
                      const double *thresholds, gpirt_outputs *out, gpirt_summary *summary,
                      int32_t options);

/* ---- checkpoint / resume --------------------------------------------------------------
 * Everything a chain needs to continue exactly where it stopped (theta, f, f*, the
 * whitened beta / threshold parameters and their derived copies, the sweep counter that keys
 * the random streams, the current jitter, the diagnostics counters) as one opaque host
 * blob.  Restoring it into a sampler created with the SAME configuration and data continues
 * the chain bit for bit.  The running sums of the summaries are not part of it.
 * No counterpart in the reference, whose scripts only save the R workspace after a finished
 * run (TAPS/gpirt_TAPS.R:41, gpirt_esm.R:86). */
int64_t gpirt_sampler_checkpoint_bytes(gpirt_sampler *s);
int gpirt_sampler_checkpoint(gpirt_sampler *s, void *blob, int64_t bytes);
int gpirt_sampler_restore(gpirt_sampler *s, const void *blob, int64_t bytes);

/* The HIP stream every kernel of this sampler is launched on (as a void*). */
void *gpirt_sampler_stream(gpirt_sampler *s);
void gpirt_sampler_destroy(gpirt_sampler *s);

/* ---- deterministic kernels with injected randomness (parity-test entry points) --------
 * Host pointers in, host pointers out; each call uploads, runs the same device
 * kernels gpirt_sampler_sweep uses, and downloads. */

/* K = sf^2 exp(-(x_i-x_j)^2 / (2 ls^2)) + jitter*I   (covSEiso at main.m:29,67;
 * SEKernel getprob_gpirt.R:41-51).  x[n] -> K[n,n] full symmetric. */
int gpirt_k_build(const double *x, int32_t n, double ls, double sf, double jitter, double *K);
/* Lower Cholesky factor of S[n,n] (chol inside mvnrnd, utils/sample_fs.m:2;
 * chol(K) idiom generate_data.R:56).  L[n,n], strict upper triangle zeroed.
 * Returns GPIRT_ERR_NOT_PD when a pivot is not positive. */
int gpirt_potrf(const double *S, int32_t n, double *L);
/* gpirt_potrf for callers that cannot allocate device memory (tight serving loops, harnesses
 * that forbid hipMalloc).  It allocates nothing, but it does copy the status word to the host and
 * waits for the stream before it returns, so it is NOT for a stream under HIP graph capture.
 * No counterpart in the reference.
 * DEVICE pointers in and out: d_S [n,n] column-major with leading dimension lds (both
 * triangles are read as gpirt_potrf reads them), d_L [n,n] with leading dimension ldl receives
 * the lower factor with a zero strict upper triangle (d_L may be d_S).  `stream` is a
 * hipStream_t (NULL: the default stream); the call returns after the stream has drained, with
 * GPIRT_ERR_NOT_PD when a pivot is not positive.  The padded matrix, the inverted diagonal
 * blocks and the status word live in one static device array of 12.6 MB per GPU that the library
 * holds until the process ends; n is limited to GPIRT_POTRF_NOALLOC_MAX_N and larger sizes are
 * refused with GPIRT_ERR_BAD_ARG.  Every call on a GPU shares that GPU's array: calls that
 * overlap on one GPU (two host threads, two streams) race and must be serialised by the caller. */
#define GPIRT_POTRF_NOALLOC_MAX_N 1024
int gpirt_potrf_noalloc(const double *d_S, int64_t lds, int32_t n, double *d_L, int64_t ldl,
                        void *stream);
/* N = L * Z for all columns at once (nu = prior'*randn, utils/elliptical_slice.m:64).
 * L[n,n] lower, Z[n,k] -> N[n,k]. */
int gpirt_trmm_lz(const double *L, const double *Z, int32_t n, int32_t k, double *N);
/* X = S^-1 B given the Cholesky factor (the solve inside gp(...), main.m:57-58).
 * L[n,n], B[n,k] -> X[n,k]. */
int gpirt_potrs(const double *L, const double *B, int32_t n, int32_t k, double *X);
/* Ordinal-probit log-likelihood (log_like_fn.m:3-4 generalised per
 * gpirt-sythetic.R:159-167).  g[n,u] latent incl. mean, y[n,u] int8, b[C+1,u]
 * thresholds (ends ignored) -> per-unit sums ll_unit[u] and per-row sums ll_row[n]
 * (either may be NULL). */
int gpirt_ordinal_loglik(const double *g, const int8_t *y, const double *b, int32_t n,
                         int32_t u, int32_t C, double *ll_unit, double *ll_row);
/* One elliptical-slice update per unit with injected randomness
 * (utils/elliptical_slice.m:47-102).  f[n,u] in/out, nu[n,u], mu[n,u] fixed mean added
 * inside the likelihood, y[n,u], b[C+1,u], unif[max_rounds+2,u] (row 0: slice height,
 * row 1: first angle, rows 2..: shrink draws).  rounds[u] out = likelihood evaluations
 * used (negative when capped), ll[u] out = log-likelihood of the returned f. */
int gpirt_ess_f(double *f, const double *nu, const double *mu, const int8_t *y,
                const double *b, const double *unif, int32_t n, int32_t u, int32_t C,
                int32_t max_rounds, int32_t *rounds, double *ll);
/* f* = K(xs, x) S^-1 f on the grid (gp(...) posterior mean, main.m:57-58).
 * x[n], L[n,n] factor of K(x,x)+jitter, f[n,u] -> fstar[n_grid,u]. */
int gpirt_fstar_predict(const double *x, const double *L, const double *f, int32_t n,
                        int32_t u, double ls, double sf, int32_t n_grid, double grid_lo,
                        double grid_step, double *fstar);
/* The counter-based generator both backends share: `count` standard normals /
 * uniforms(0,1) of stream (seed, tag, sweep, unit). */
int gpirt_rng_normal(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                     int64_t count, double *out);
int gpirt_rng_uniform(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                      int64_t count, double *out);

#ifdef __cplusplus
}
#endif
#endif /* GPIRT_H */
