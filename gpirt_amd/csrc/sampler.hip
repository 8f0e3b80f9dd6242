// Host side of the C ABI declared in include/gpirt.h: the resident sampler (one Gibbs
// sweep = main.m:46-87), gpirt_run (one chain of gpirtMCMC, gpirt-sythetic.R:88-94) and the
// parity entry points that expose the same device kernels with injected inputs.
#include "kernels.h"

#include <dlfcn.h>

#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

namespace gpirt {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
}

// ---- roctx ranges ---------------------------------------------------------------------------
// With GPIRT_ROCTX=1 every sub-step of a sweep is bracketed by a roctx range (k_build+potrf,
// nu+ess_f, ess_param, fstar, theta, snap), so that a `rocprofv3 --marker-trace --kernel-trace`
// run can be cut by sub-step.  The marker library is looked up at run time: nothing links
// against it, and without the variable (or the library) the ranges cost one branch.
struct Roctx {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
  Roctx() {
    const char *e = getenv("GPIRT_ROCTX");
    if (!e || !*e || *e == '0') return;
    for (const char *name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4", "libroctx64.so"}) {
      void *h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) continue;
      push = (int (*)(const char *))dlsym(h, "roctxRangePushA");
      pop = (int (*)())dlsym(h, "roctxRangePop");
      if (push && pop) return;
      push = nullptr; pop = nullptr;
    }
  }
};
static const Roctx &roctx() { static Roctx r; return r; }
struct Range {
  bool on;
  explicit Range(const char *name) : on(roctx().push != nullptr) { if (on) roctx().push(name); }
  ~Range() { if (on) roctx().pop(); }
};

// ---- small RAII helpers for the entry points ---------------------------------------------
struct DevBuf {
  void *p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  int alloc(size_t bytes, bool zero = true) {
    GP_HIP(hipMalloc(&p, bytes ? bytes : 8));
    if (zero) GP_HIP(hipMemset(p, 0, bytes ? bytes : 8));
    return GPIRT_OK;
  }
  template <typename T> T *as() const { return static_cast<T *>(p); }
};

static int have_device() {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess || c < 1) {
    set_error("no HIP device visible: the gfx950 kernels cannot run (there is no CPU fallback)");
    return GPIRT_ERR_NO_DEVICE;
  }
  return GPIRT_OK;
}

// host column-major [rows, cols] -> device padded [ld, cols_pad] (zero pad)
static int upload_padded(const double *h, int rows, int cols, double *d, int64_t ld, hipStream_t st) {
  GP_HIP(hipMemcpy2DAsync(d, sizeof(double) * ld, h, sizeof(double) * rows, sizeof(double) * rows,
                          cols, hipMemcpyHostToDevice, st));
  return GPIRT_OK;
}
static int download_padded(double *h, int rows, int cols, const double *d, int64_t ld, hipStream_t st) {
  GP_HIP(hipMemcpy2DAsync(h, sizeof(double) * rows, d, sizeof(double) * ld, sizeof(double) * rows,
                          cols, hipMemcpyDeviceToHost, st));
  return GPIRT_OK;
}

static double temporal_k(const gpirt_config &c, double d) {
  if (c.theta_os == 0.0) return (c.theta_ls < 0.0) ? 1.0 : (d == 0.0 ? 1.0 : 0.0);
  double os2 = c.theta_os * c.theta_os;
  if (c.theta_ls <= 0.0) return os2;
  double a = fabs(d) / c.theta_ls;
  if (c.kernel == GPIRT_KERNEL_MATERN52) {
    double q = 2.2360679774997896964 * a;
    return os2 * (1.0 + q + q * q / 3.0) * exp(-q);
  }
  return os2 * exp(-0.5 * a * a);
}

static int host_chol_lower(std::vector<double> &A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[j + (size_t)n * j];
    for (int k = 0; k < j; ++k) d -= A[j + (size_t)n * k] * A[j + (size_t)n * k];
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    A[j + (size_t)n * j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = A[i + (size_t)n * j];
      for (int k = 0; k < j; ++k) s -= A[i + (size_t)n * k] * A[j + (size_t)n * k];
      A[i + (size_t)n * j] = s / d;
    }
    for (int i = 0; i < j; ++i) A[i + (size_t)n * j] = 0.0;
  }
  return 0;
}

}  // namespace gpirt

using namespace gpirt;

enum { PH_KBUILD = 0, PH_POTRF, PH_NU, PH_ESSF, PH_PARAM, PH_FSTAR, PH_THETA, PH_SNAP, PH_Z, PH_COUNT };

struct gpirt_sampler {
  gpirt_config cfg;
  int n, m, H, C, P, Np, Np_pad, mloc, ML, U_pad, G, G_pad, j0, nb, fixed_thr, device;
  uint64_t seed;
  uint32_t sweep;
  double jitter_now;
  hipStream_t st;
  int8_t *d_y, *d_yT, *d_yTl = nullptr;   // yTl: yT with a chunk's items lane-major (bucketed respondent step)
  double *d_theta, *d_f, *d_Z, *d_nu, *d_S, *d_inv, *d_fwork, *d_xtmp, *d_W, *d_Gt;
  double *d_ebeta, *d_delta, *d_beta, *d_bthr, *d_betaS, *d_bthrS, *d_bpm, *d_bps;
  double *d_xc, *d_tnu, *d_prop, *d_hh, *d_phi, *d_pmin, *d_pmax, *d_buf, *d_Lt, *d_tpm, *d_tps;
  double *d_llu;
  double *d_onehot, *d_T, *d_lp;   // static (grid) respondent step only
  double *d_tab;                   // slice respondent step, table variant: [U_pad, C, G] (else null)
  double *d_nuparts;               // partial products of nu = L Z when K is split (else null)
  int nu_splits;
  double *d_part;                  // bucketed evaluation of the table variant (H = 1, else null)
  int32_t *d_gidx, *d_order, *d_bstart, *d_bcur, *d_ghist;
  int th_chunk, th_nchunks;
  int th_slot0, th_nslots;         // GPIRT_FIXED_ORDER_SUM: this rank's first block among all ranks' blocks
  double *d_sum_th, *d_sum_th2, *d_sum_p, *d_sum_arg;   // running sums of the summaries (lazy)
  double *d_nodes, *d_lag;   // Chebyshev nodes on the grid's range and the interpolation matrix
  // low-rank route of the f* step (see setup_fstar_lowrank): barycentric weights, B [R,128],
  // B^T Lag [128,G_pad]; per sweep Phi [(Np_pad+128),128] per group, its Gram matrix and factor,
  // X = M^-1 B^T Lag, and the slabs of the Gram product
  double *d_bw, *d_B, *d_BL, *d_feat, *d_Mm, *d_minv, *d_X, *d_gparts;
  int fs_lowrank;            // the route is set up
  // generator form of the prior factor (GPIRT_GENERATOR_PRIOR, see phase_factor_generator):
  // per 128-row block b the factors R_b of J_b and L_bb of the block's Schur complement with
  // their inverses, T_b, L_bb^-1 T_b, W_b^T, and the per-block partial products of nu
  double *d_Jb, *d_jinv, *d_Tb, *d_Sbb, *d_linv, *d_Et, *d_Wt, *d_V;
  int gen_prior;             // the route is set up
  int gen_now;               // this sweep's factor is in generator form
  int super_inv_ready;       // the off-diagonal blocks of this sweep's inverted super-blocks are built
  int theta_inside;          // this sweep's inputs all lie inside the nodes' range
  int h_info2;               // host copy of the status of this sweep's low-rank f* factorisation
  int fs_nodes, fs_rank;     // nodes in use and their padded count (0: exact cross-covariance)
  int summary_draws, summary_theta_draws;
  int32_t *d_done, *d_info, *d_remaining;   // d_remaining = d_info + 4: one block, one copy per look
  int32_t *h_look;                           // pinned host mirror of that block
  unsigned long long *d_counters;  // [3][2]: f, param, theta x (evaluations, capped); [6] see ThetaArgs::ll_seen
  int64_t theta_rounds;
  int jitter_retries;
  double phase_ms[PH_COUNT];
  int64_t phase_cnt[PH_COUNT];
  std::vector<hipEvent_t> ev;
  size_t ev_used;
  std::vector<int> ev_phase;
  // the factorisation's rank-512 trailing updates, timed one by one (event pairs around each launch)
  std::vector<hipEvent_t> upd_ev;
  size_t upd_used;
  double upd_ms, upd_flop, upd_flop_pending;
  int64_t upd_cnt;
  // nu = L Z beside the factorisation: the rows of the first nu_overlap super-panels (SUP rows
  // each) are computed on a second stream, in chunks of nu_chunk tile rows, once those rows of L
  // are final and the trailing update that followed them has drained -- beside the next
  // super-panels' chain of small kernels, which leaves most CUs idle.  A chunk puts one workgroup
  // on most CUs and leaves some CUs empty: the diagonal-block kernel needs a whole CU.
  // Look-ahead (launch_potrf with a side stream): the bulk of each trailing update runs on the
  // second stream beside the next super-panel's chain.
  // Either leaves the chain the same bit for bit (same per-tile arithmetic), and which
  // combination is fastest depends on the share of the items this rank holds, so the sampler
  // times the factor + nu span of every combination it may use in its first sweeps (two sweeps
  // each, the second one timed) and keeps the fastest.
  hipStream_t st2;
  // The slice steps in two shares of the units: the second share's kernels go to a stream of lower priority, so that
  // its workgroups fill the chip while the first share's last, longest units finish (a launch of these kernels ends
  // with ~0.2 ms in which a few units run on a mostly empty chip), and the f* step, which needs f only, runs beside
  // the second share's parameter update.  share_units: units in the second share (0: one launch per step);
  // share_now: what the sweep under way uses until its shares are joined; share_last: what the last sweep used.
  hipStream_t st_lo = nullptr;
  hipEvent_t ev_share[3] = {nullptr, nullptr, nullptr};   // nu ready / second share's f done / its parameters done
  int share_units = 0, share_now = 0, share_last = 0;
  int lookahead;       // look-ahead may be used (1), is forced (2)
  int nu_lone;         // the chunks of nu beside the factorisation run one workgroup to a CU
  std::vector<hipEvent_t> ev_ahead;
  std::vector<hipEvent_t> ev_panel;
  hipEvent_t ev_join, ev_z, ev_pref, ev_tail, ev_span[2][2];   // [sweep parity][begin, end] of the factor + nu span
  int pref_on_st2;     // this sweep's input-only factors of the low-rank f* step went to the second stream
  int nu_overlap, nu_chunk, nu_rows_done, z_filled;
  int nu_tail;         // super-panels of L whose share of the LAST rows of nu is added beside the factorisation
  int nu_overlap_la, nu_tail_la;   // the two numbers for the schedules with look-ahead
  int nu_tail_done;    // ... of them, enqueued in this sweep so far
  int ov_now;          // this sweep computes rows of nu beside the factorisation
  int la_now;          // this sweep factors with look-ahead
  int sched_cand[4], sched_ncand;   // schedules to choose from: bit 0 rows of nu beside, bit 1 look-ahead
  int sched_choice;    // -1: still measuring; else the schedule in use
  int dense_sweeps;    // sweeps that took the dense route so far
  int span_open;       // parity + 1 of the span this sweep has begun (0: none)
  int span_kind[2];    // the candidate whose span of that parity was measured (-1: none)
  float sched_ms[4];   // measured span per candidate (the shorter of its two timed sweeps)
  int span_last[2];    // that span was the candidate's last timed sweep
};

// units in the second share of the slice steps: three eighths from 1024 units on, or what GPIRT_SLICE_SHARES says
static void choose_slice_shares(gpirt_sampler *s) {
  s->share_units = 0;
  if (!s->st_lo) return;
  const char *e = getenv("GPIRT_SLICE_SHARES");
  // (one rank only by default: the gain is half a percent, and a schedule with a third stream has not run beside a collective)
  int k = s->mloc * s->P >= 1024 && s->cfg.world <= 1 ? (3 * s->U_pad) / 8 : 0;
  if (e) k = atoi(e);
  s->share_units = std::max(0, std::min(k, s->U_pad - 1));
}

static void sampler_free(gpirt_sampler *s) {
  if (!s) return;
  void *ptrs[] = {s->d_y, s->d_yT, s->d_yTl, s->d_theta, s->d_f, s->d_Z, s->d_nu, s->d_S, s->d_inv, s->d_fwork, s->d_xtmp, s->d_W,
                  s->d_Gt, s->d_ebeta, s->d_delta, s->d_beta, s->d_bthr, s->d_betaS, s->d_bthrS,
                  s->d_bpm, s->d_bps, s->d_xc, s->d_tnu, s->d_prop, s->d_hh, s->d_phi, s->d_pmin,
                  s->d_pmax, s->d_buf, s->d_Lt, s->d_tpm, s->d_tps, s->d_llu, s->d_done, s->d_info,
                  s->d_counters, s->d_onehot, s->d_T, s->d_lp,
                  s->d_sum_th, s->d_sum_th2, s->d_sum_p, s->d_sum_arg, s->d_nodes, s->d_lag, s->d_tab,
                  s->d_part, s->d_gidx, s->d_order, s->d_bstart, s->d_bcur, s->d_ghist, s->d_nuparts,
                  s->d_bw, s->d_B, s->d_BL, s->d_feat, s->d_Mm, s->d_minv, s->d_X, s->d_gparts,
                  s->d_Jb, s->d_jinv, s->d_Tb, s->d_Sbb, s->d_linv, s->d_Et, s->d_Wt, s->d_V};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  if (s->h_look) (void)hipHostFree(s->h_look);
  for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->upd_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_panel) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_ahead) (void)hipEventDestroy(e);

  if (s->ev_join) (void)hipEventDestroy(s->ev_join);
  if (s->ev_z) (void)hipEventDestroy(s->ev_z);
  if (s->ev_pref) (void)hipEventDestroy(s->ev_pref);
  if (s->ev_tail) (void)hipEventDestroy(s->ev_tail);
  for (int a = 0; a < 2; ++a) for (int b = 0; b < 2; ++b) if (s->ev_span[a][b]) (void)hipEventDestroy(s->ev_span[a][b]);
  for (hipEvent_t e : s->ev_share) if (e) (void)hipEventDestroy(e);
  if (s->st_lo) (void)hipStreamDestroy(s->st_lo);
  if (s->st2) (void)hipStreamDestroy(s->st2);
  if (s->st) (void)hipStreamDestroy(s->st);
  delete s;
}

template <typename T>
static int dalloc(T *&p, size_t count) {
  GP_HIP(hipMalloc((void **)&p, (count ? count : 1) * sizeof(T)));
  GP_HIP(hipMemset(p, 0, (count ? count : 1) * sizeof(T)));
  return GPIRT_OK;
}

static void params_host(const gpirt_sampler *s, int jl, const double *e, const double *dl,
                        double *beta, double *b, const double *bpm, const double *bps) {
  for (int k = 0; k < 3; ++k) beta[k] = bpm[k + 3 * jl] + bps[k + 3 * jl] * e[k];
  int C = s->C;
  b[0] = -INFINITY; b[C] = INFINITY;
  b[1] = dl[0];
  for (int c = 2; c < C; ++c) b[c] = b[c - 1] + exp(dl[c - 1]);
}

// The f* step needs K(x, xs) S^-1 F on the G = 1001 grid points.  As a function of the grid
// point, sf^2 exp(-(x - s)^2 / (2 ls^2)) is entire, so it is reproduced to rounding error by
// polynomial interpolation from R Chebyshev nodes on the grid's range:
//     K(x, xs) = K(x, c) Lag,   Lag[j, g] = l_j(xs_g)   (barycentric Lagrange basis),
// and the solve needs R (128) right-hand sides instead of G (1024).  R is accepted only if the
// interpolation error, measured here on a spread of x, is below 1e-13 sf^2; otherwise (short
// length scales) the step falls back to the exact cross-covariance.
static int setup_fstar_interpolation(gpirt_sampler *s) {
  const int G = s->G, G_pad = s->G_pad;
  const double lo = s->cfg.grid_lo, hi = lo + s->cfg.grid_step * (G - 1), ls = s->cfg.irf_ls;
  const double kPi = 3.14159265358979323846;
  for (int R = TILE; R <= 2 * TILE && R < G; R += TILE) {
    std::vector<double> c(R), w(R);
    for (int j = 0; j < R; ++j) {
      c[j] = 0.5 * (lo + hi) + 0.5 * (hi - lo) * cos(kPi * j / (R - 1));
      w[j] = (j & 1) ? -1.0 : 1.0;
    }
    w[0] *= 0.5; w[R - 1] *= 0.5;
    std::vector<double> lag((size_t)R * G_pad, 0.0);          // [R, G_pad], node index contiguous
    for (int g = 0; g < G; ++g) {
      const double xg = lo + s->cfg.grid_step * g;
      int hit = -1;
      double den = 0.0;
      for (int j = 0; j < R; ++j) {
        if (xg == c[j]) { hit = j; break; }
        den += w[j] / (xg - c[j]);
      }
      for (int j = 0; j < R; ++j)
        lag[j + (size_t)R * g] = hit >= 0 ? (j == hit ? 1.0 : 0.0) : (w[j] / (xg - c[j])) / den;
    }
    double err = 0.0;
    for (int t = 0; t <= 240 && err <= 1e-13; ++t) {
      const double x = (lo - 3.0) + (hi - lo + 6.0) * t / 240.0;
      std::vector<double> kc(R);
      for (int j = 0; j < R; ++j) kc[j] = exp(-0.5 * (c[j] - x) * (c[j] - x) / (ls * ls));
      for (int g = 0; g < G; g += 7) {
        double acc = 0.0;
        for (int j = 0; j < R; ++j) acc += kc[j] * lag[j + (size_t)R * g];
        const double xg = lo + s->cfg.grid_step * g;
        err = fmax(err, fabs(acc - exp(-0.5 * (xg - x) * (xg - x) / (ls * ls))));
      }
    }
    if (err > 1e-13) continue;
    GP_TRY(dalloc(s->d_nodes, (size_t)R));
    GP_TRY(dalloc(s->d_lag, (size_t)R * G_pad));
    GP_HIP(hipMemcpy(s->d_nodes, c.data(), sizeof(double) * R, hipMemcpyHostToDevice));
    GP_HIP(hipMemcpy(s->d_lag, lag.data(), sizeof(double) * lag.size(), hipMemcpyHostToDevice));
    s->fs_nodes = R; s->fs_rank = R;
    return GPIRT_OK;
  }
  s->fs_rank = 0;
  return GPIRT_OK;
}

// Eigen-decomposition of a symmetric matrix by cyclic Jacobi rotations (host, set-up only):
// A [n,n] is overwritten, its diagonal ends as the eigenvalues, V's columns as the vectors.
static void jacobi_eigen(std::vector<double> &A, int n, std::vector<double> &V) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[i + (size_t)n * i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int p = 0; p < n; ++p) {
      diag += A[p + (size_t)n * p] * A[p + (size_t)n * p];
      for (int q = p + 1; q < n; ++q) off += A[p + (size_t)n * q] * A[p + (size_t)n * q];
    }
    if (off <= 1e-40 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = A[p + (size_t)n * q];
        if (apq == 0.0) continue;
        const double app = A[p + (size_t)n * p], aqq = A[q + (size_t)n * q];
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + t * t), sn = t * c;
        for (int k = 0; k < n; ++k) {     // columns p, q
          const double akp = A[k + (size_t)n * p], akq = A[k + (size_t)n * q];
          A[k + (size_t)n * p] = c * akp - sn * akq;
          A[k + (size_t)n * q] = sn * akp + c * akq;
        }
        for (int k = 0; k < n; ++k) {     // rows p, q
          const double apk = A[p + (size_t)n * k], aqk = A[q + (size_t)n * k];
          A[p + (size_t)n * k] = c * apk - sn * aqk;
          A[q + (size_t)n * k] = sn * apk + c * aqk;
        }
        for (int k = 0; k < n; ++k) {
          const double vkp = V[k + (size_t)n * p], vkq = V[k + (size_t)n * q];
          V[k + (size_t)n * p] = c * vkp - sn * vkq;
          V[k + (size_t)n * q] = sn * vkp + c * vkq;
        }
      }
  }
}

// Low-rank route of the f* step.  With l(x) the Lagrange basis of the Chebyshev nodes c and
// Kcc = K(c, c), the kernel is reproduced to rounding level in BOTH arguments,
//   K(x, x') = l(x)^T Kcc l(x'),
// so with Kcc = V D V^T (eigenvalues below 1e-17 of the largest dropped), B = V D^1/2 and
// Phi = P B (P = the basis at the inputs):  S = K(x,x) + s2 I = Phi Phi^T + s2 I,
// K(xs, x) = (B^T Lag)^T Phi^T, and by the push-through identity
//   f*^T = F^T S^-1 K(x, xs) = (F^T Phi) (s2 I + Phi^T Phi)^-1 (B^T Lag):
// one [units x n x 128] product and a 128 x 128 factorisation replace the two triangular
// solves with the n x n factor.  Everything that depends only on the nodes is built here.
static int setup_fstar_lowrank(gpirt_sampler *s) {
  s->fs_lowrank = 0;
  const int R = s->fs_rank, G = s->G, G_pad = s->G_pad;
  if (R <= 0) return GPIRT_OK;
  const double lo = s->cfg.grid_lo, hi = lo + s->cfg.grid_step * (G - 1);
  const double ls = s->cfg.irf_ls, sf2 = s->cfg.irf_sf * s->cfg.irf_sf;
  std::vector<double> c(R), w(R), lag((size_t)R * G_pad);
  GP_HIP(hipMemcpy(c.data(), s->d_nodes, sizeof(double) * R, hipMemcpyDeviceToHost));
  GP_HIP(hipMemcpy(lag.data(), s->d_lag, sizeof(double) * lag.size(), hipMemcpyDeviceToHost));
  for (int j = 0; j < R; ++j) w[j] = (j & 1) ? -1.0 : 1.0;
  w[0] *= 0.5; w[R - 1] *= 0.5;
  std::vector<double> Kcc((size_t)R * R), V;
  for (int a = 0; a < R; ++a)
    for (int b = 0; b < R; ++b) Kcc[a + (size_t)R * b] = sf2 * exp(-0.5 * (c[a] - c[b]) * (c[a] - c[b]) / (ls * ls));
  std::vector<double> A = Kcc;
  jacobi_eigen(A, R, V);
  double dmax = 0.0;
  for (int k = 0; k < R; ++k) dmax = fmax(dmax, A[k + (size_t)R * k]);
  std::vector<int> keep;
  for (int k = 0; k < R; ++k)
    if (A[k + (size_t)R * k] > 1e-17 * dmax) keep.push_back(k);
  if ((int)keep.size() > TILE) return GPIRT_OK;            // kernel too rough for 128 columns
  std::vector<double> B((size_t)R * TILE, 0.0);             // [R, 128], zero columns beyond the rank
  for (size_t q = 0; q < keep.size(); ++q) {
    const double sd = sqrt(A[keep[q] + (size_t)R * keep[q]]);
    for (int a = 0; a < R; ++a) B[a + (size_t)R * q] = V[a + (size_t)R * keep[q]] * sd;
  }
  // check: l(x)^T B B^T l(x') against the kernel on a lattice of the range
  double err = 0.0;
  const int T = 48;
  std::vector<double> PB((size_t)T * TILE);
  std::vector<double> xt(T);
  for (int t = 0; t < T; ++t) {
    xt[t] = lo + (hi - lo) * (t + 0.37) / T;
    std::vector<double> l(R);
    double den = 0.0;
    for (int j = 0; j < R; ++j) den += w[j] / (xt[t] - c[j]);
    for (int j = 0; j < R; ++j) l[j] = (w[j] / (xt[t] - c[j])) / den;
    for (int q = 0; q < TILE; ++q) {
      double acc = 0.0;
      for (int j = 0; j < R; ++j) acc += l[j] * B[j + (size_t)R * q];
      PB[t + (size_t)T * q] = acc;
    }
  }
  for (int t = 0; t < T; ++t)
    for (int u = 0; u < T; ++u) {
      double acc = 0.0;
      for (int q = 0; q < TILE; ++q) acc += PB[t + (size_t)T * q] * PB[u + (size_t)T * q];
      err = fmax(err, fabs(acc - sf2 * exp(-0.5 * (xt[t] - xt[u]) * (xt[t] - xt[u]) / (ls * ls))));
    }
  if (err > 1e-12 * sf2) return GPIRT_OK;
  std::vector<double> BL((size_t)TILE * G_pad, 0.0);        // [128, G_pad] = B^T Lag
  for (int g = 0; g < G; ++g)
    for (int q = 0; q < (int)keep.size(); ++q) {
      double acc = 0.0;
      for (int j = 0; j < R; ++j) acc += B[j + (size_t)R * q] * lag[j + (size_t)R * g];
      BL[q + (size_t)TILE * g] = acc;
    }
  const int64_t ldp = s->Np_pad + TILE;
  GP_TRY(dalloc(s->d_bw, (size_t)R));
  GP_TRY(dalloc(s->d_B, B.size()));
  GP_TRY(dalloc(s->d_BL, BL.size()));
  GP_TRY(dalloc(s->d_feat, (size_t)ldp * TILE * s->P));
  GP_TRY(dalloc(s->d_Mm, (size_t)TILE * TILE * s->P));
  GP_TRY(dalloc(s->d_minv, (size_t)inv_group_elems(TILE) * s->P));
  GP_TRY(dalloc(s->d_X, (size_t)TILE * G_pad * s->P));
  GP_TRY(dalloc(s->d_gparts, (size_t)(ldp / TILE) * TILE * TILE));
  GP_HIP(hipMemcpy(s->d_bw, w.data(), sizeof(double) * R, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(s->d_B, B.data(), sizeof(double) * B.size(), hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(s->d_BL, BL.data(), sizeof(double) * BL.size(), hipMemcpyHostToDevice));
  GP_HIP(hipMemset(s->d_feat, 0, sizeof(double) * (size_t)ldp * TILE * s->P));
  GP_HIP(hipMemset(s->d_minv, 0, sizeof(double) * (size_t)inv_group_elems(TILE) * s->P));
  s->fs_lowrank = 1;
  return GPIRT_OK;
}

// Buffers of the generator form of the prior factor (one IRF group only).
static int setup_generator_prior(gpirt_sampler *s) {
  s->gen_prior = 0;
  if (!s->fs_lowrank || s->P != 1) return GPIRT_OK;
  const size_t nblk = (size_t)s->Np_pad / TILE, blk = (size_t)TILE * TILE;
  GP_TRY(dalloc(s->d_Jb, nblk * blk));
  GP_TRY(dalloc(s->d_jinv, nblk * (size_t)inv_group_elems(TILE)));
  GP_TRY(dalloc(s->d_Tb, nblk * blk));
  GP_TRY(dalloc(s->d_Sbb, nblk * blk));
  GP_TRY(dalloc(s->d_linv, nblk * (size_t)inv_group_elems(TILE)));
  GP_TRY(dalloc(s->d_Et, nblk * blk));
  GP_TRY(dalloc(s->d_Wt, nblk * blk));
  GP_TRY(dalloc(s->d_V, (size_t)s->Np_pad * s->ML));
  s->gen_prior = 1;
  return GPIRT_OK;
}

static int check_cfg(const gpirt_config *cfg) {
  if (!cfg) { set_error("cfg is NULL"); return GPIRT_ERR_BAD_ARG; }
  if (cfg->struct_size != (int32_t)sizeof(gpirt_config)) {
    set_error("gpirt_config.struct_size %d != %zu (header mismatch)", cfg->struct_size,
              sizeof(gpirt_config));
    return GPIRT_ERR_BAD_ARG;
  }
  if (cfg->n < 1 || cfg->m < 1 || cfg->H < 1 || cfg->H > MAXH || cfg->C < 2 || cfg->C > MAXC) {
    set_error("bad shape n=%d m=%d H=%d (max %d) C=%d (2..%d)", cfg->n, cfg->m, cfg->H, MAXH, cfg->C,
              MAXC);
    return GPIRT_ERR_BAD_ARG;
  }
  if (cfg->n_grid < 2 || !(cfg->grid_step > 0) || !(cfg->irf_ls > 0) || !(cfg->jitter >= 0)) {
    set_error("bad grid / kernel hyper-parameters");
    return GPIRT_ERR_BAD_ARG;
  }
  if (cfg->theta_mode != GPIRT_THETA_ESS && cfg->theta_mode != GPIRT_THETA_GRID) {
    set_error("unknown theta_mode %d", cfg->theta_mode);
    return GPIRT_ERR_BAD_ARG;
  }
  if (cfg->theta_mode == GPIRT_THETA_GRID && cfg->H != 1) {
    set_error("GPIRT_THETA_GRID is the static respondent step: H must be 1 (got %d)", cfg->H);
    return GPIRT_ERR_BAD_ARG;
  }
  if ((cfg->flags & GPIRT_GENERATOR_PRIOR) && (cfg->flags & (GPIRT_EXACT_FSTAR | GPIRT_FSTAR_SOLVE))) {
    set_error("GPIRT_GENERATOR_PRIOR keeps no dense factor: it cannot be combined with the f* routes "
              "that solve with it (GPIRT_EXACT_FSTAR, GPIRT_FSTAR_SOLVE)");
    return GPIRT_ERR_BAD_ARG;
  }
  if (cfg->world > 1 && !cfg->allreduce) {
    set_error("world=%d needs an allreduce hook", cfg->world);
    return GPIRT_ERR_BAD_ARG;
  }
  if (cfg->item_end > cfg->item_begin && (cfg->item_begin < 0 || cfg->item_end > cfg->m)) {
    set_error("item range [%d,%d) outside [0,%d)", cfg->item_begin, cfg->item_end, cfg->m);
    return GPIRT_ERR_BAD_ARG;
  }
  return GPIRT_OK;
}

extern "C" {

const char *gpirt_version(void) { return "gpirt-hip 0.1 (gfx950)"; }
const char *gpirt_last_error(void) { return g_err; }

int gpirt_device_count(void) {
  int c = 0;
  if (hipGetDeviceCount(&c) != hipSuccess) return 0;
  return c;
}

int gpirt_set_device(int device) {
  GP_TRY(have_device());
  GP_HIP(hipSetDevice(device));
  return GPIRT_OK;
}

int gpirt_device_copy(void *dst, const void *src, int64_t bytes, int32_t to_device) {
  GP_TRY(have_device());
  if (!dst || !src || bytes < 0) { set_error("gpirt_device_copy: bad argument"); return GPIRT_ERR_BAD_ARG; }
  GP_HIP(hipMemcpy(dst, src, (size_t)bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
  return GPIRT_OK;
}

void gpirt_config_defaults(gpirt_config *c) {
  memset(c, 0, sizeof *c);
  c->struct_size = (int32_t)sizeof *c;
  c->H = 1; c->C = 2;
  c->sample_iterations = 500; c->burn_iterations = 500; c->thin = 1;
  c->seed = 1;
  c->theta_os = 1.0; c->theta_ls = 1.0;
  c->kernel = GPIRT_KERNEL_RBF;
  c->jitter = 1e-6; c->irf_ls = 1.0; c->irf_sf = 1.0;
  c->n_grid = 1001; c->grid_lo = -5.0; c->grid_step = 0.01;
  c->theta_mode = GPIRT_THETA_ESS;
  c->max_ess_rounds = 200;
  c->world = 1;
}

int gpirt_sampler_create(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
                         const double *bpm_in, const double *bps_in, const double *tpm_in,
                         const double *tps_in, const double *thresholds, gpirt_sampler **out) {
  GP_TRY(check_cfg(cfg));
  if (!y || !out) { set_error("y/out is NULL"); return GPIRT_ERR_BAD_ARG; }
  GP_TRY(have_device());
  gpirt_sampler *s = new gpirt_sampler();  // value-initialised: every scalar and pointer is 0
  s->cfg = *cfg;
  if (s->cfg.max_ess_rounds <= 0) s->cfg.max_ess_rounds = 200;
  s->n = cfg->n; s->m = cfg->m; s->H = cfg->H; s->C = cfg->C; s->G = cfg->n_grid;
  s->P = cfg->constant_irf ? 1 : cfg->H;
  s->Np = cfg->constant_irf ? cfg->n * cfg->H : cfg->n;
  s->j0 = cfg->item_end > cfg->item_begin ? cfg->item_begin : 0;
  int j1 = cfg->item_end > cfg->item_begin ? cfg->item_end : cfg->m;
  s->mloc = j1 - s->j0;
  s->ML = (int)pad_up(s->mloc);
  s->U_pad = s->ML * s->P;
  s->Np_pad = (int)pad_up(s->Np);
  s->G_pad = (int)pad_up(s->G);
  s->nb = s->Np_pad / TILE;
  s->seed = cfg->seed + 0x9E3779B97F4A7C15ull * (uint64_t)cfg->chain;
  s->jitter_now = cfg->jitter;
  s->fixed_thr = thresholds != nullptr;
  const int n = s->n, m = s->m, H = s->H, C = s->C, P = s->P, Np_pad = s->Np_pad,
            mloc = s->mloc, ML = s->ML, U_pad = s->U_pad, G_pad = s->G_pad;
  int rc = GPIRT_OK;
  auto fail = [&](int code) { sampler_free(s); return code; };
  if (hipGetDevice(&s->device) != hipSuccess) return fail(GPIRT_ERR_HIP);
  if (hipStreamCreate(&s->st) != hipSuccess) {
    set_error("stream creation failed");
    return fail(GPIRT_ERR_HIP);
  }

#define A_(ptr, count) if ((rc = dalloc(ptr, (size_t)(count)))) return fail(rc)
  A_(s->d_y, (size_t)Np_pad * U_pad);
  A_(s->d_yT, (size_t)ML * H * n);
  A_(s->d_theta, (size_t)n * H);
  A_(s->d_f, (size_t)Np_pad * U_pad);
  A_(s->d_Z, (size_t)Np_pad * U_pad);
  // nu doubles as the [U_pad, R] image of F^T S^-1 K(x, c) in the f* step (R <= 2 TILE nodes)
  A_(s->d_nu, (size_t)max(Np_pad, 2 * TILE) * U_pad);
  A_(s->d_S, (size_t)Np_pad * Np_pad * P);
  A_(s->d_inv, (size_t)inv_group_elems(Np_pad) * P);
  A_(s->d_fwork, (size_t)potrf_work_elems(Np_pad, P));
  A_(s->d_xtmp, (size_t)4 * SUP * G_pad * P);
  A_(s->d_W, (size_t)Np_pad * G_pad * P);
  A_(s->d_Gt, (size_t)U_pad * G_pad);
  A_(s->d_ebeta, 3 * (size_t)U_pad);
  A_(s->d_delta, (size_t)(C - 1) * U_pad);
  A_(s->d_beta, 3 * (size_t)U_pad);
  A_(s->d_bthr, (size_t)(C + 1) * U_pad);
  A_(s->d_betaS, 3 * (size_t)U_pad);
  A_(s->d_bthrS, (size_t)(C + 1) * U_pad);
  A_(s->d_bpm, 3 * (size_t)mloc);
  A_(s->d_bps, 3 * (size_t)mloc);
  A_(s->d_xc, (size_t)n * H);
  A_(s->d_tnu, (size_t)n * H);
  A_(s->d_prop, (size_t)THETA_SPEC * n * H);
  A_(s->d_hh, n); A_(s->d_phi, (size_t)THETA_SPEC * n); A_(s->d_pmin, n); A_(s->d_pmax, n);
  A_(s->d_buf, (size_t)THETA_SPEC * n);
  A_(s->d_Lt, (size_t)H * H);
  A_(s->d_tpm, 2 * (size_t)n); A_(s->d_tps, 2 * (size_t)n);
  A_(s->d_llu, U_pad);
  A_(s->d_done, n); A_(s->d_info, (size_t)(4 + s->cfg.max_ess_rounds + 4));
  s->d_remaining = s->d_info + 4;
  if (hipHostMalloc((void **)&s->h_look, sizeof(int32_t) * (size_t)(4 + s->cfg.max_ess_rounds + 4)) != hipSuccess) {
    set_error("pinned host buffer: allocation failed");
    return fail(GPIRT_ERR_HIP);
  }
  memset(s->h_look, 0, sizeof(int32_t) * (size_t)(4 + s->cfg.max_ess_rounds + 4));
  A_(s->d_counters, 8);   // [6]: checksum of the respondent step's log-likelihood sums (GPIRT_FIXED_ORDER_SUM)
  // nu = L Z with few column tiles (a rank's share of the items, or a handful of items): the
  // longest row panel alone would set the time and most CUs would hold one workgroup or none,
  // so K is cut into slabs that run side by side.  Measured at n = 10 000: 512 columns
  // 1.62 -> 1.02 ms with 8 slabs (1.20 with 4), 1024 columns 2.11 -> 1.88 ms with 4 (2 brings
  // nothing), 2048 columns and more: no gain.
  s->nu_splits = 1;
  if (P == 1 && Np_pad >= 1024) {
    const int tiles = (Np_pad / TILE) * (ML / TILE);
    int want = tiles < 512 ? 8 : (tiles < 1024 ? 4 : 1);
    while (want > 1 && (Np_pad / 16) % want) want /= 2;   // slabs: multiples of the GEMM's depth step
    s->nu_splits = want;
    if (want > 1) A_(s->d_nuparts, (size_t)want * Np_pad * ML);
  }
  s->nu_overlap = 0; s->nu_overlap_la = 0; s->nu_tail_la = 0;
  // look-ahead in the factorisation: one IRF group, from 4096 inputs on; GPIRT_LOOKAHEAD=0 / 1
  // takes it out of / forces it into the schedule
  const char *la_env = getenv("GPIRT_LOOKAHEAD");
  s->lookahead = P == 1 && Np_pad >= 8 * SUP && !(la_env && atoi(la_env) == 0) ? (la_env ? 2 : 1) : 0;
  s->nu_lone = !(getenv("GPIRT_NU_LONE") && atoi(getenv("GPIRT_NU_LONE")) == 0);
  int ov_forced = 0;
  if (s->nu_splits > 1 && Np_pad >= 8 * SUP && !(cfg->flags & GPIRT_NO_NU_OVERLAP)) {
    // nu in slabs of K (few column tiles): the rows of each finished super-panel go to the second stream
    // as a launch of the same slabs -- same sums, same order of adding them -- on the fixed grid;
    // GPIRT_NU_OVERLAP=k: the first k super-panels' rows (0: none)
    const int nsup = inv_nsup(Np_pad);
    const char *e = getenv("GPIRT_NU_OVERLAP"), *c = getenv("GPIRT_NU_CHUNK");
    // (measured at n = 10 000 with look-ahead: 512 items 86.4 -> 90.4 sweeps/s, 1024 items 78.0 -> 80.8 with 18 of 20
    // super-panels in chunks of 12 tile rows; chunks of 4 or 8 tile rows cost the factorisation more than they save)
    s->nu_overlap = max(0, min(e ? atoi(e) : (9 * nsup) / 10, nsup));
    ov_forced = e != nullptr;
    s->nu_chunk = c ? max(1, atoi(c)) : 12;
    s->nu_tail = 0;
    s->nu_overlap_la = s->nu_overlap; s->nu_tail_la = 0;
  }
  if (s->nu_splits == 1 && Np_pad >= 8 * SUP && !(cfg->flags & GPIRT_NO_NU_OVERLAP)) {
    // GPIRT_NU_OVERLAP=k forces k super-panels (0: none) and GPIRT_NU_CHUNK=t the chunk size;
    // by default the rows left for the launch after the factorisation make at most 512 tiles
    // (one round of workgroups: those rows have the longest K, a tile of them runs ~2 ms)
    const int nsup = inv_nsup(Np_pad), ncol = ML / TILE;
    const char *e = getenv("GPIRT_NU_OVERLAP"), *c = getenv("GPIRT_NU_CHUNK");
    const char *tl = getenv("GPIRT_NU_TAIL");
    const bool tail = !(tl && atoi(tl) == 0);
    int k = (s->nb - 512 / ncol + 3) / 4;
    k = max((nsup + 1) / 2, min(k, nsup - 2));
    // with the last rows' early panels taken beside the factorisation too, fewer super-panels of rows: one fewer with
    // the factorisation's step-by-step form, two fewer since its steps take the diagonal super-block first (the chain
    // of small kernels is shorter and the chunks lag more: n = 10 000, m = 4096: 15 super-panels + 14 panels 56.2,
    // 14 + 13 57.0, 14 + 12 57.3, 13 + 12 54.5, 16 + 15 53.3 sweeps/s)
    if (tail && k > 3) k -= 2;
    s->nu_overlap = max(0, e ? min(atoi(e), nsup) : k);
    ov_forced = e != nullptr;
    s->nu_chunk = c ? max(1, atoi(c)) : max(1, min(16, 224 / ncol));
    // The first chain phases have next to nothing beside them: the rows of nu that are final by
    // then are few and short.  The rows left for the launch after the factorisation can take
    // there, panel by panel, the share of each finished 512-column panel of L:
    // nu[tail] += L[tail, panel] Z[panel] in panel order, the accumulators started from nu, which
    // is the single launch's sum in the single launch's order.  GPIRT_NU_TAIL = number of panels.
    // (all panels before the last overlapped super-panel by default; measured at n = 10 000, m = 4096:
    // 51.8 -> 53.7 sweeps/s with 15 super-panels of rows and 14 panels, the factor + nu span 14.7 -> 14.1 ms)
    s->nu_tail = s->nu_overlap > 0 ? max(0, min(tl ? atoi(tl) : s->nu_overlap - 1, s->nu_overlap - 1)) : 0;
    // With look-ahead the second stream carries the bulk of every trailing update already: there the
    // panels' shares of the last rows cost more than they give, and more rows beside the late,
    // emptier chain phases are better (n = 10 000, 2048 items: 11 super-panels + 10 panels 64.2, 16-17
    // super-panels + none 68.6 sweeps/s, look-ahead alone 67.0)
    s->nu_overlap_la = e ? s->nu_overlap : min(nsup, max(s->nu_overlap, (17 * nsup) / 20));
    s->nu_tail_la = tl ? s->nu_tail : 0;
  }
  // the schedules this sampler may use, plainest first
  s->sched_ncand = 0;
  for (int c = 0; c < 4; ++c) {
    const bool ov = c & 1, la = c & 2;
    if (ov ? s->nu_overlap == 0 : (ov_forced && s->nu_overlap > 0)) continue;
    if (la ? s->lookahead == 0 : s->lookahead == 2) continue;
    s->sched_cand[s->sched_ncand++] = c;
  }
  s->sched_choice = s->sched_ncand == 1 ? s->sched_cand[0] : -1;
  if (s->nu_overlap > 0 || s->lookahead) {
    bool ok = hipStreamCreateWithFlags(&s->st2, hipStreamNonBlocking) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_join, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_z, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_pref, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&s->ev_tail, hipEventDisableTiming) == hipSuccess;
    for (int a = 0; a < 2 && ok; ++a)
      for (int b = 0; b < 2 && ok; ++b) ok = hipEventCreate(&s->ev_span[a][b]) == hipSuccess;
    for (int i = 0; i < max(s->nu_overlap, s->nu_overlap_la) && ok; ++i) {
      hipEvent_t ev;
      ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
      if (ok) s->ev_panel.push_back(ev);
    }
    for (int i = 0; s->lookahead && i < 2 * inv_nsup(Np_pad) && ok; ++i) {
      hipEvent_t ev;
      ok = hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
      if (ok) s->ev_ahead.push_back(ev);
    }
    if (!ok) { set_error("second stream: creation failed"); return fail(GPIRT_ERR_HIP); }
  }
  // the slice steps in two shares of the units (see st_lo), from 1024 units of 1024 rows on: n = 10 000, 4096 items,
  // 60.4 -> 60.85 sweeps/s with 1536 units in the second share (1024 ... 2048 units: 60.6-60.85); n = 5000, 2048 items
  // 218 -> 224; n = 2500, 1024 items 481 -> 494 (there it is the f* step beside the second share that gains).  GPIRT_SLICE_SHARES=k: k units in the second share (0: one launch per step); read again at
  // every call that runs sweeps.
  if (s->Np >= 1024 && mloc * P >= 1024) {
    int least = 0, greatest = 0;
    bool ok = hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess &&
              hipStreamCreateWithPriority(&s->st_lo, hipStreamNonBlocking, least) == hipSuccess;
    for (int i = 0; i < 3 && ok; ++i) ok = hipEventCreateWithFlags(&s->ev_share[i], hipEventDisableTiming) == hipSuccess;
    if (!ok) { set_error("low-priority stream: creation failed"); return fail(GPIRT_ERR_HIP); }
  }
  choose_slice_shares(s);
  s->span_kind[0] = s->span_kind[1] = -1;
  for (int c = 0; c < 4; ++c) s->sched_ms[c] = 1e30f;
  if (cfg->theta_mode == GPIRT_THETA_GRID) {
    if (cfg->flags & GPIRT_FIXED_ORDER_SUM) {
      set_error("GPIRT_FIXED_ORDER_SUM is for the slice variant of the respondent step (theta_mode GPIRT_THETA_ESS)");
      return fail(GPIRT_ERR_BAD_ARG);
    }
    A_(s->d_onehot, (size_t)C * ML * Np_pad);
    A_(s->d_T, (size_t)G_pad * C * ML);
    A_(s->d_lp, (size_t)G_pad * Np_pad);
  } else {
    // a respondent's likelihood is a function of the grid index: from about a thousand
    // respondents on, tabulating it once per sweep is cheaper than evaluating cells per round
    const bool table = (cfg->flags & GPIRT_THETA_TABLE) ||
                       (n >= 1024 && !(cfg->flags & GPIRT_THETA_DIRECT));
    if (table) A_(s->d_tab, (size_t)U_pad * C * s->G);
    const bool fixed = (cfg->flags & GPIRT_FIXED_ORDER_SUM) != 0;
    if (fixed && !(table && H == 1 && s->G <= THETA_BUCKET_MAX_G)) {
      set_error("GPIRT_FIXED_ORDER_SUM needs H = 1, the tabulated respondent step and a grid of at most %d points",
                THETA_BUCKET_MAX_G);
      return fail(GPIRT_ERR_BAD_ARG);
    }
    if (table && H == 1 && s->G <= THETA_BUCKET_MAX_G) {
      s->th_chunk = theta_bucket_chunk(C);
      if (fixed) {
        // blocks of the WHOLE item set, the same for every number of ranks
        s->th_chunk = min(s->th_chunk, GPIRT_SUM_BLOCK);
        if (s->j0 % s->th_chunk || ((s->j0 + mloc) % s->th_chunk && s->j0 + mloc != s->m)) {
          set_error("GPIRT_FIXED_ORDER_SUM: items [%d, %d) do not begin and end on multiples of %d", s->j0,
                    s->j0 + mloc, s->th_chunk);
          return fail(GPIRT_ERR_BAD_ARG);
        }
      }
      s->th_nchunks = (mloc + s->th_chunk - 1) / s->th_chunk;
      s->th_slot0 = fixed ? s->j0 / s->th_chunk : 0;
      s->th_nslots = fixed ? (s->m + s->th_chunk - 1) / s->th_chunk : s->th_nchunks;
      A_(s->d_part, (size_t)s->th_nslots * THETA_SPEC * n);
      A_(s->d_gidx, (size_t)THETA_SPEC * n); A_(s->d_order, (size_t)THETA_SPEC * n); A_(s->d_bstart, (size_t)s->G + 1); A_(s->d_bcur, s->G); A_(s->d_ghist, s->G);
      // the wide chunks read the responses from a lane-major copy (whole chunks only: ML is a multiple of 128)
      if ((s->th_chunk == 512 || s->th_chunk == 1024) && ML % s->th_chunk == 0) A_(s->d_yTl, (size_t)ML * n);
    }
  }
#undef A_

  // ---- host staging -------------------------------------------------------------------
  std::vector<int8_t> hy((size_t)Np_pad * U_pad, 0), hyT((size_t)ML * H * n, 0);
  for (int h = 0; h < H; ++h)
    for (int jl = 0; jl < mloc; ++jl)
      for (int i = 0; i < n; ++i) {
        int8_t v = y[i + (size_t)n * ((s->j0 + jl) + (size_t)m * h)];
        if (v < 0 || v > C) v = 0;
        size_t u = cfg->constant_irf ? (size_t)jl : (size_t)jl + (size_t)ML * h;
        size_t r = cfg->constant_irf ? (size_t)i + (size_t)n * h : (size_t)i;
        hy[r + (size_t)Np_pad * u] = v;
        hyT[(size_t)ML * H * i + (size_t)ML * h + jl] = v;
      }
  std::vector<double> hth((size_t)n * H);
  for (int h = 0; h < H; ++h)
    for (int i = 0; i < n; ++i)
      hth[i + (size_t)n * h] = theta_init ? theta_init[i + (size_t)n * h]
                                          : rng_normal_at(s->seed, TAG_INIT, 0, (uint32_t)i, 0);
  std::vector<double> hbpm(3 * (size_t)mloc), hbps(3 * (size_t)mloc);
  for (int jl = 0; jl < mloc; ++jl)
    for (int k = 0; k < 3; ++k) {
      hbpm[k + 3 * jl] = bpm_in ? bpm_in[k + 3 * (size_t)(s->j0 + jl)] : 0.0;
      hbps[k + 3 * jl] = bps_in ? bps_in[k + 3 * (size_t)(s->j0 + jl)] : 1.0;
    }
  std::vector<double> heb(3 * (size_t)U_pad, 0.0), hdl((size_t)(C - 1) * U_pad, 0.0),
      hbeta(3 * (size_t)U_pad, 0.0), hb((size_t)(C + 1) * U_pad, 0.0),
      hbetaS(3 * (size_t)U_pad, 0.0), hbS((size_t)(C + 1) * U_pad, 0.0);
  for (int p = 0; p < P; ++p)
    for (int jl = 0; jl < mloc; ++jl) {
      size_t u = (size_t)jl + (size_t)ML * p;
      double b[MAXC + 2], beta[3];
      for (int c = 1; c < C; ++c)
        b[c] = thresholds ? thresholds[c] : (C == 2 ? 0.0 : -1.5 + 3.0 * (double)(c - 1) / (double)(C - 2));
      double *dl = &hdl[(size_t)(C - 1) * u];
      dl[0] = b[1];
      for (int c = 2; c < C; ++c) dl[c - 1] = log(b[c] - b[c - 1]);
      params_host(s, jl, &heb[3 * u], dl, beta, b, hbpm.data(), hbps.data());
      for (int k = 0; k < 3; ++k) { hbeta[k + 3 * u] = beta[k]; hbetaS[u + (size_t)U_pad * k] = beta[k]; }
      for (int c = 0; c <= C; ++c) { hb[c + (size_t)(C + 1) * u] = b[c]; hbS[u + (size_t)U_pad * c] = b[c]; }
    }
  std::vector<double> hLt((size_t)H * H);
  for (int a = 0; a < H; ++a)
    for (int b2 = 0; b2 < H; ++b2)
      hLt[a + (size_t)H * b2] = temporal_k(*cfg, (double)(a - b2)) + (a == b2 ? 1e-6 : 0.0);
  if (host_chol_lower(hLt, H)) {
    set_error("temporal kernel (theta_os=%g, theta_ls=%g) is not positive definite", cfg->theta_os,
              cfg->theta_ls);
    return fail(GPIRT_ERR_NOT_PD);
  }
  std::vector<double> htpm(2 * (size_t)n, 0.0), htps(2 * (size_t)n, 0.0);
  if (tpm_in) memcpy(htpm.data(), tpm_in, sizeof(double) * 2 * n);
  if (tps_in) memcpy(htps.data(), tps_in, sizeof(double) * 2 * n);

#define UP_(dst, vec) \
  if (hipMemcpy(dst, (vec).data(), (vec).size() * sizeof((vec)[0]), hipMemcpyHostToDevice) != hipSuccess) { \
    set_error("upload of " #vec " failed"); return fail(GPIRT_ERR_HIP); }
  UP_(s->d_y, hy); UP_(s->d_yT, hyT); UP_(s->d_theta, hth); UP_(s->d_bpm, hbpm); UP_(s->d_bps, hbps);
  UP_(s->d_ebeta, heb); UP_(s->d_delta, hdl); UP_(s->d_beta, hbeta); UP_(s->d_bthr, hb);
  UP_(s->d_betaS, hbetaS); UP_(s->d_bthrS, hbS); UP_(s->d_Lt, hLt); UP_(s->d_tpm, htpm);
  UP_(s->d_tps, htps);
#undef UP_
  if (!(cfg->flags & GPIRT_EXACT_FSTAR)) {
    if ((rc = setup_fstar_interpolation(s))) return fail(rc);
    if (!(cfg->flags & GPIRT_FSTAR_SOLVE) && (rc = setup_fstar_lowrank(s))) return fail(rc);
    if ((cfg->flags & GPIRT_GENERATOR_PRIOR) && (rc = setup_generator_prior(s))) return fail(rc);
  }
  if (s->d_yTl) {
    if ((rc = launch_theta_lane_major(s->st, s->d_yT, n, ML, s->th_chunk, s->d_yTl))) return fail(rc);
    if (hipStreamSynchronize(s->st) != hipSuccess) { set_error("lane-major copy of the responses failed"); return fail(GPIRT_ERR_HIP); }
  }
  if (cfg->theta_mode == GPIRT_THETA_GRID) {
    if ((rc = launch_grid_onehot(s->st, s->d_yT, n, mloc, ML, C, s->d_onehot))) return fail(rc);
    if (hipStreamSynchronize(s->st) != hipSuccess) { set_error("one-hot coding failed"); return fail(GPIRT_ERR_HIP); }
  }
  *out = s;
  return GPIRT_OK;
}

}  // extern "C"

// ---- phases -----------------------------------------------------------------------------
static int mark(gpirt_sampler *s, int phase) {
  if (s->ev_used == s->ev.size()) {
    hipEvent_t e;
    GP_HIP(hipEventCreate(&e));
    s->ev.push_back(e);
    s->ev_phase.push_back(0);
  }
  s->ev_phase[s->ev_used] = phase;
  GP_HIP(hipEventRecord(s->ev[s->ev_used], s->st));
  s->ev_used++;
  return GPIRT_OK;
}

// events are recorded as (begin of phase k); the time of phase k is the gap to the next mark
static int collect_times(gpirt_sampler *s) {
  for (size_t i = 0; i + 1 < s->ev_used; ++i) {
    int ph = s->ev_phase[i];
    if (ph < 0) continue;
    float ms = 0;
    GP_HIP(hipEventElapsedTime(&ms, s->ev[i], s->ev[i + 1]));
    s->phase_ms[ph] += ms;
    s->phase_cnt[ph] += 1;
  }
  s->ev_used = 0;
  for (size_t i = 0; i + 1 < s->upd_used; i += 2) {
    float ms = 0;
    GP_HIP(hipEventElapsedTime(&ms, s->upd_ev[i], s->upd_ev[i + 1]));
    s->upd_ms += ms;
    s->upd_cnt += 1;
  }
  s->upd_flop += s->upd_flop_pending;
  s->upd_flop_pending = 0.0;
  s->upd_used = 0;
  return GPIRT_OK;
}

// brackets one trailing update of the factorisation with a pair of events (see PotrfHook)
static int update_timer(void *ctx, int begin, double flop, hipStream_t stream) {
  gpirt_sampler *s = (gpirt_sampler *)ctx;
  if (s->upd_used == s->upd_ev.size()) {
    hipEvent_t e;
    GP_HIP(hipEventCreate(&e));
    s->upd_ev.push_back(e);
  }
  GP_HIP(hipEventRecord(s->upd_ev[s->upd_used++], stream));
  if (!begin) s->upd_flop_pending += flop;
  return GPIRT_OK;
}

// Generator form of the prior factor.  With S = Phi Phi^T + s2 I (setup_fstar_lowrank) the
// Cholesky factor is block semiseparable: for the 128-row blocks b, with G_b the Gram matrix
// of the rows of Phi before block b and J_b = s2 I + G_b = R_b R_b^T,
//   T_b = s Phi_b R_b^-T,   L_bb L_bb^T = s2 I + T_b T_b^T   (the block's Schur complement),
//   L_ib = Phi_i W_b for every later row i,   W_b^T = s (L_bb^-1 T_b) R_b^-1,
// so that  nu = L z  is  nu_b = L_bb z_b + Phi_b sum_{b' < b} W_b' z_b'  (chol(K)'*z of
// sample_fs.m:2, demo.m:45, without the n x n matrix).  All blocks are independent given the
// prefix sums: the two factorisations run as batches of the diagonal kernel.  *done = 0 when
// an input lies outside the nodes' range; the caller then takes the dense route for this sweep.
static int fstar_status(gpirt_sampler *s);
static int phase_factor_generator(gpirt_sampler *s, int *done) {
  *done = 0;
  GP_TRY(mark(s, PH_KBUILD));
  // ([2], the status of the low-rank f* step, stays: see fstar_status)
  GP_HIP(hipMemsetAsync(s->d_info, 0, 2 * sizeof(int), s->st));
  GP_TRY(launch_bary(s->st, s->d_theta, s->Np, s->Np, s->Np_pad, s->P, s->d_nodes, s->d_bw, s->fs_nodes,
                     s->cfg.grid_lo, s->cfg.grid_lo + s->cfg.grid_step * (s->G - 1), s->d_W,
                     (int64_t)s->Np_pad * s->G_pad, s->d_info + 1));
  GP_HIP(hipMemcpyAsync(s->h_look, s->d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, s->st));
  GP_HIP(hipStreamSynchronize(s->st));
  if (s->fs_lowrank) { s->h_info2 = s->h_look[2]; GP_TRY(fstar_status(s)); }
  if (s->h_look[1]) return GPIRT_OK;
  s->theta_inside = 1;
  GP_TRY(mark(s, PH_POTRF));
  const int nblk = s->Np_pad / TILE;
  const int64_t ldp = s->Np_pad + TILE, blk = (int64_t)TILE * TILE, sI = inv_group_elems(TILE);
  const double s2 = s->jitter_now, sd = sqrt(s2);
  GemmArgs a{};   // Phi = P B
  a.A = s->d_W; a.lda = s->Np_pad; a.B = s->d_B; a.ldb = s->fs_nodes;
  a.C = s->d_feat; a.ldc = ldp;
  a.M = s->Np_pad; a.N = TILE; a.K = s->fs_nodes; a.batch = 1; a.alpha = 1.0; a.beta = 0.0;
  GP_TRY(launch_gemm(s->st, false, false, a));
  GemmArgs m{};   // Gram matrix of every block's rows
  m.A = s->d_feat; m.lda = ldp; m.strideA = TILE; m.B = m.A; m.ldb = ldp; m.strideB = TILE;
  m.C = s->d_gparts; m.ldc = TILE; m.strideC = blk;
  m.M = TILE; m.N = TILE; m.K = TILE; m.batch = nblk; m.alpha = 1.0; m.beta = 0.0;
  GP_TRY(launch_gemm(s->st, true, false, m));
  GP_TRY(launch_prefix_gram(s->st, s->d_gparts, nblk, s2, s->d_Jb));
  GP_TRY(launch_potrf(s->st, s->d_Jb, TILE, nblk, s->d_jinv, s->d_info, s->d_fwork));
  GemmArgs t{};   // T_b = s Phi_b R_b^-T
  t.A = s->d_feat; t.lda = ldp; t.strideA = TILE; t.B = s->d_jinv; t.ldb = SUP; t.strideB = sI;
  t.C = s->d_Tb; t.ldc = TILE; t.strideC = blk;
  t.M = TILE; t.N = TILE; t.K = TILE; t.batch = nblk; t.alpha = sd; t.beta = 0.0;
  GP_TRY(launch_gemm(s->st, false, true, t));
  GP_TRY(launch_scaled_identity(s->st, s->d_Sbb, TILE, 0, nblk, blk, s2));
  GemmArgs c{};   // s2 I + T_b T_b^T
  c.A = s->d_Tb; c.lda = TILE; c.strideA = blk; c.B = s->d_Tb; c.ldb = TILE; c.strideB = blk;
  c.C = s->d_Sbb; c.ldc = TILE; c.strideC = blk;
  c.M = TILE; c.N = TILE; c.K = TILE; c.batch = nblk; c.alpha = 1.0; c.beta = 1.0;
  GP_TRY(launch_gemm(s->st, false, true, c));
  GP_TRY(launch_potrf(s->st, s->d_Sbb, TILE, nblk, s->d_linv, s->d_info, s->d_fwork));
  GP_TRY(launch_tril(s->st, s->d_Sbb, nblk));
  GemmArgs e{};   // L_bb^-1 T_b
  e.A = s->d_linv; e.lda = SUP; e.strideA = sI; e.B = s->d_Tb; e.ldb = TILE; e.strideB = blk;
  e.C = s->d_Et; e.ldc = TILE; e.strideC = blk;
  e.M = TILE; e.N = TILE; e.K = TILE; e.batch = nblk; e.alpha = 1.0; e.beta = 0.0;
  GP_TRY(launch_gemm(s->st, false, false, e));
  GemmArgs w{};   // W_b^T = s (L_bb^-1 T_b) R_b^-1
  w.A = s->d_Et; w.lda = TILE; w.strideA = blk; w.B = s->d_jinv; w.ldb = SUP; w.strideB = sI;
  w.C = s->d_Wt; w.ldc = TILE; w.strideC = blk;
  w.M = TILE; w.N = TILE; w.K = TILE; w.batch = nblk; w.alpha = sd; w.beta = 0.0;
  GP_TRY(launch_gemm(s->st, false, false, w));
  GP_HIP(hipMemcpyAsync(s->h_look, s->d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, s->st));
  GP_HIP(hipStreamSynchronize(s->st));
  const int info = s->h_look[0];
  // a block that is not positive definite: this sweep takes the dense route, which escalates
  // the jitter and counts the retries
  if (info) return GPIRT_OK;
  *done = 1;
  return GPIRT_OK;
}

// nu = L z for every unit at once (sample_fs.m:2), rows [r0, r1) of it
static GemmArgs nu_args(const gpirt_sampler *s, int r0, int r1) {
  GemmArgs g{};
  g.A = s->d_S + r0; g.lda = s->Np_pad; g.strideA = (int64_t)s->Np_pad * s->Np_pad;
  g.B = s->d_Z; g.ldb = s->Np_pad; g.strideB = (int64_t)s->Np_pad * s->ML;
  g.C = s->d_nu + r0; g.ldc = s->Np_pad; g.strideC = (int64_t)s->Np_pad * s->ML;
  g.M = r1 - r0; g.N = s->ML; g.K = r1; g.batch = s->P; g.alpha = 1.0; g.beta = 0.0;
  g.k_trim = 1; g.k_base = r0;
  return g;
}

// The same rows with K cut into nu_splits slabs that run side by side (the GEMM's batch index), each
// into its own copy of nu; launch_sum_slabs adds the copies in slab order.  Slabs that begin at or
// beyond row r1 hold nothing for these rows and are left out (adding their zeros changes no bit).
static int nu_split_launch(gpirt_sampler *s, hipStream_t q, int r0, int r1, int lone) {
  if (r1 <= r0) return GPIRT_OK;
  GemmArgs g = nu_args(s, r0, r1);
  const int kc = s->Np_pad / s->nu_splits, nsl = min(s->nu_splits, (r1 + kc - 1) / kc);
  g.K = kc; g.k_shift = kc; g.batch = nsl;
  g.strideA = (int64_t)kc * s->Np_pad; g.strideB = kc;
  g.C = s->d_nuparts + r0;
  g.k_trim = r0 > 0 || r1 < s->Np_pad ? 4 : 1;   // (a kernel name of its own in a profile)
  g.lone = lone;
  GP_TRY(launch_gemm(q, false, false, g));
  return launch_sum_slabs(q, s->d_nuparts + r0, nsl, g.strideC, r1 - r0, s->ML, s->Np_pad, s->d_nu + r0, s->Np_pad);
}

// Called by the factorisation when block columns [o, oe) are final and the trailing update
// behind them is enqueued: whole chunks of final rows go to the second stream.
static int nu_panel_hook(void *ctx, int o, int oe) {
  gpirt_sampler *s = (gpirt_sampler *)ctx;
  const int nov = s->la_now ? s->nu_overlap_la : s->nu_overlap, ntail = s->la_now ? s->nu_tail_la : s->nu_tail;
  const int sp = o / 4, lim = min(nov * 4, s->nb);
  if (sp >= nov) return GPIRT_OK;
  const int avail = min(oe, lim);
  bool recorded = false;
  for (int t = s->nu_rows_done / TILE; t + s->nu_chunk <= avail || (avail == lim && t < lim);) {
    const int t1 = min(t + s->nu_chunk, avail);
    if (!recorded) {
      GP_HIP(hipEventRecord(s->ev_panel[sp], s->st));
      GP_HIP(hipStreamWaitEvent(s->st2, s->ev_panel[sp], 0));
      recorded = true;
    }
    if (s->nu_splits > 1) {
      GP_TRY(nu_split_launch(s, s->st2, t * TILE, t1 * TILE, s->nu_lone));
    } else {
      GemmArgs c = nu_args(s, t * TILE, t1 * TILE);
      c.lone = s->nu_lone;
      GP_TRY(launch_gemm(s->st2, false, false, c));
    }
    s->nu_rows_done = t1 * TILE;
    t = t1;
  }
  if (sp < ntail && sp == s->nu_tail_done && lim < s->nb) {
    if (!recorded) {
      GP_HIP(hipEventRecord(s->ev_panel[sp], s->st));
      GP_HIP(hipStreamWaitEvent(s->st2, s->ev_panel[sp], 0));
    }
    const int r0 = lim * TILE, k0 = sp * SUP;
    GemmArgs g{};
    g.A = s->d_S + r0 + (int64_t)s->Np_pad * k0; g.lda = s->Np_pad;
    g.B = s->d_Z + k0; g.ldb = s->Np_pad;
    g.C = s->d_nu + r0; g.ldc = s->Np_pad;
    g.M = s->Np_pad - r0; g.N = s->ML; g.K = SUP; g.batch = 1; g.alpha = 1.0; g.beta = sp ? 1.0 : 0.0;
    g.lone = s->nu_lone;
    GP_TRY(launch_gemm(s->st2, false, false, g));
    // the launch after the factorisation adds to these rows on the main stream: it waits for this event
    // (recorded ahead of the chunks that this stream receives later, so those still run beside that launch)
    GP_HIP(hipEventRecord(s->ev_tail, s->st2));
    s->nu_tail_done = sp + 1;
  }
  return GPIRT_OK;
}

// The sweep before this one is measured from the begin of its factorisation to the begin of this
// sweep's, which has been waited for by now.  Candidate c runs in the dense route's sweeps 3c (warming up), 3c + 1 and
// 3c + 2 (timed, the shorter span counts: one timed sweep per candidate let a box's jitter pick);
// once every candidate has its time the fastest stays, a plainer one unless the other is 1 % faster.
constexpr int SCHED_SWEEPS = 3;
static int schedule_tune(gpirt_sampler *s) {
  if (s->sched_choice >= 0) return GPIRT_OK;
  const int prev = (s->dense_sweeps + 1) & 1, kind = s->span_kind[prev];
  if (kind >= 0) {
    float ms = 0;
    // the whole sweep, from its factorisation's begin to this sweep's: what runs beside the factorisation
    // also moves the phases after it (the input-only factors of the f* step on the second stream)
    GP_HIP(hipEventElapsedTime(&ms, s->ev_span[prev][0], s->ev_span[s->dense_sweeps & 1][0]));
    s->sched_ms[kind] = fminf(s->sched_ms[kind], ms);
    s->span_kind[prev] = -1;
    if (kind == s->sched_ncand - 1 && s->span_last[prev]) {
      int best = 0;
      for (int c = 1; c < s->sched_ncand; ++c)
        if (s->sched_ms[c] < 0.99f * s->sched_ms[best]) best = c;
      s->sched_choice = s->sched_cand[best];
      if (getenv("GPIRT_SCHED_DEBUG"))
        for (int c = 0; c < s->sched_ncand; ++c)
          fprintf(stderr, "gpirt schedule %d (rows of nu beside %d, look-ahead %d): %.3f ms%s\n", s->sched_cand[c],
                  s->sched_cand[c] & 1, (s->sched_cand[c] >> 1) & 1, s->sched_ms[c], c == best ? "  <- kept" : "");
    }
  }
  return GPIRT_OK;
}

static int fill_z(gpirt_sampler *s) {
  GP_TRY(mark(s, PH_Z));
  GP_TRY(launch_fill_normal(s->st, s->d_Z, s->Np_pad, s->Np, s->Np_pad, s->U_pad, s->mloc, s->ML,
                            s->m, s->j0, s->seed, s->sweep));
  s->z_filled = 1;
  return GPIRT_OK;
}

static int phase_factor(gpirt_sampler *s) {
  Range range("gpirt:factor (k_build + potrf)");
  s->gen_now = 0;
  s->z_filled = 0;
  s->nu_rows_done = 0;
  s->nu_tail_done = 0;
  s->span_open = 0;
  s->pref_on_st2 = 0;
  if (s->gen_prior) {
    int done = 0;
    GP_TRY(phase_factor_generator(s, &done));
    if (done) { s->gen_now = 1; return GPIRT_OK; }
  }
  const int par = s->dense_sweeps & 1;
  int sched = s->sched_choice;
  if (sched < 0) {
    const int ci = min(s->dense_sweeps / SCHED_SWEEPS, s->sched_ncand - 1);
    sched = s->sched_cand[ci];
    GP_HIP(hipEventRecord(s->ev_span[par][0], s->st));
    s->span_kind[par] = s->dense_sweeps % SCHED_SWEEPS >= 1 && s->dense_sweeps / SCHED_SWEEPS < s->sched_ncand ? ci : -1;
    s->span_last[par] = s->dense_sweeps % SCHED_SWEEPS == SCHED_SWEEPS - 1;
    s->span_open = par + 1;
  }
  s->ov_now = sched & 1;
  s->la_now = (sched & 2) != 0;
  PotrfHook hook{nullptr, s, update_timer, nullptr, nullptr};
  if (s->la_now) { hook.side = s->st2; hook.ev = s->ev_ahead.data(); }
  for (int attempt = 0;; ++attempt) {
    s->nu_rows_done = 0;
    s->nu_tail_done = 0;
    if (attempt && s->st2) {
      // a failed attempt's launches on the second stream still read and write the matrix
      GP_HIP(hipEventRecord(s->ev_join, s->st2));
      GP_HIP(hipStreamWaitEvent(s->st, s->ev_join, 0));
    }
    GP_TRY(mark(s, PH_KBUILD));
    GP_TRY(launch_k_build(s->st, s->d_theta, s->Np, s->Np, s->Np_pad, s->P, s->cfg.irf_ls,
                          s->cfg.irf_sf, s->jitter_now, s->d_S, false));
    if (s->ov_now && !s->z_filled) {
      // the normal draws depend on nothing but the sweep counter: they run on the second stream,
      // ahead of the chunks that read them, beside the first super-panel's chain of small kernels
      // (beside the kernel-matrix build they would share its HBM bandwidth); the join event is
      // free here, and the wait keeps them behind the previous sweep's launches that read Z
      GP_HIP(hipEventRecord(s->ev_join, s->st));
      GP_HIP(hipStreamWaitEvent(s->st2, s->ev_join, 0));
      GP_TRY(launch_fill_normal(s->st2, s->d_Z, s->Np_pad, s->Np, s->Np_pad, s->U_pad, s->mloc, s->ML,
                                s->m, s->j0, s->seed, s->sweep));
      GP_HIP(hipEventRecord(s->ev_z, s->st2));
      s->z_filled = 1;
    }
    GP_TRY(mark(s, PH_POTRF));
    // info[0]: pivot of a failed factorisation; [1]: an input outside the range of the f* nodes
    // ([2], the status of the low-rank f* step's factorisation, is sticky: only fstar_status clears it,
    // after it has reported the failure, so a sweep that nobody looked behind cannot lose it)
    GP_HIP(hipMemsetAsync(s->d_info, 0, 2 * sizeof(int), s->st));
    if (s->fs_lowrank)   // the Lagrange basis at the inputs; it waits in W, which is free until the f* step
      GP_TRY(launch_bary(s->st, s->d_theta, s->Np, s->Np, s->Np_pad, s->P, s->d_nodes, s->d_bw, s->fs_nodes,
                         s->cfg.grid_lo, s->cfg.grid_lo + s->cfg.grid_step * (s->G - 1), s->d_W,
                         (int64_t)s->Np_pad * s->G_pad, s->d_info + 1));
    hook.fn = s->ov_now ? nu_panel_hook : nullptr;
    GP_TRY(launch_potrf(s->st, s->d_S, s->Np_pad, s->P, s->d_inv, s->d_info, s->d_fwork, &hook,
                        POTRF_DEFER_INVERSES | POTRF_SUBSTITUTION));
    s->super_inv_ready = 0;
    GP_HIP(hipMemcpyAsync(s->h_look, s->d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, s->st));
    GP_HIP(hipStreamSynchronize(s->st));
    s->theta_inside = !s->h_look[1];
    // an earlier sweep's low-rank f* step that failed with nobody looking (the grid-mode respondent step
    // never reads the status inside a sweep) ends the chain here
    if (s->fs_lowrank) { s->h_info2 = s->h_look[2]; GP_TRY(fstar_status(s)); }
    const int info = s->h_look[0];
    if (!info) {
      GP_TRY(schedule_tune(s));
      s->dense_sweeps++;
      return GPIRT_OK;
    }
    if (s->jitter_now >= 1e-2) {
      set_error("K(theta,theta)+%g I is not positive definite (pivot %d)", s->jitter_now, info);
      return GPIRT_ERR_NOT_PD;
    }
    s->jitter_now *= 10.0;
    s->jitter_retries++;
  }
}

static int fstar_lowrank_prefactor(gpirt_sampler *s, hipStream_t q);

static int phase_f(gpirt_sampler *s) {
  Range range("gpirt:f (nu = L Z + slice update)");
  if (!s->z_filled) GP_TRY(fill_z(s));
  else if (s->ov_now && !s->gen_now) GP_HIP(hipStreamWaitEvent(s->st, s->ev_z, 0));   // drawn on the second stream
  GP_TRY(mark(s, PH_NU));
  // the rows that were not computed beside the factorisation (all of them by default)
  const int r0 = s->gen_now ? 0 : s->nu_rows_done;
  GemmArgs g = nu_args(s, r0, s->Np_pad);
  if (!s->gen_now && r0 > 0 && s->nu_tail_done > 0) {
    // the first nu_tail_done panels of L are in these rows of nu already (nu_panel_hook): go on from there
    const int k0 = s->nu_tail_done * SUP;
    g.A += (int64_t)s->Np_pad * k0; g.B += k0; g.K -= k0; g.k_base -= k0; g.beta = 1.0;
    // those panels' launches ran on the second stream; the last of them wrote the rows this launch adds to
    GP_HIP(hipStreamWaitEvent(s->st, s->ev_tail, 0));
  }
  // the XCD-aware tile order keeps HBM traffic down when there are many column tiles per row
  // panel; with fewer it is slower than plain longest-first (3.83 vs 3.59 ms at 2048 columns)
  // (k_trim 4: longest-first under a kernel name of its own, so that a profile lists it apart)
  g.k_trim = r0 > 0 ? 4 : ((g.M / TILE) * (s->ML / TILE) >= 2048 ? 3 : 1);
  if (s->gen_now) {
    // nu_b = L_bb z_b + Phi_b sum_{b' < b} W_b' z_b'  (phase_factor_generator)
    const int nblk = s->Np_pad / TILE;
    const int64_t ldp = s->Np_pad + TILE, blk = (int64_t)TILE * TILE, vb = (int64_t)TILE * s->ML;
    GemmArgs v{};   // V_b = W_b z_b
    v.A = s->d_Wt; v.lda = TILE; v.strideA = blk; v.B = s->d_Z; v.ldb = s->Np_pad; v.strideB = TILE;
    v.C = s->d_V; v.ldc = TILE; v.strideC = vb;
    v.M = TILE; v.N = s->ML; v.K = TILE; v.batch = nblk; v.alpha = 1.0; v.beta = 0.0;
    GP_TRY(launch_gemm(s->st, true, false, v));
    GP_TRY(launch_prefix_blocks(s->st, s->d_V, vb, nblk));
    GemmArgs l{};   // L_bb z_b
    l.A = s->d_Sbb; l.lda = TILE; l.strideA = blk; l.B = s->d_Z; l.ldb = s->Np_pad; l.strideB = TILE;
    l.C = s->d_nu; l.ldc = s->Np_pad; l.strideC = TILE;
    l.M = TILE; l.N = s->ML; l.K = TILE; l.batch = nblk; l.alpha = 1.0; l.beta = 0.0;
    GP_TRY(launch_gemm(s->st, false, false, l));
    GemmArgs u{};   // + Phi_b U_b
    u.A = s->d_feat; u.lda = ldp; u.strideA = TILE; u.B = s->d_V; u.ldb = TILE; u.strideB = vb;
    u.C = s->d_nu; u.ldc = s->Np_pad; u.strideC = TILE;
    u.M = TILE; u.N = s->ML; u.K = TILE; u.batch = nblk; u.alpha = 1.0; u.beta = 1.0;
    GP_TRY(launch_gemm(s->st, false, false, u));
  } else if (s->nu_splits > 1) {
    // the slabs of K run side by side and a small kernel adds them in fixed order
    GP_TRY(nu_split_launch(s, s->st, r0, s->Np_pad, 0));
  } else {
    GP_TRY(launch_gemm(s->st, false, false, g));
  }
  if (s->ov_now && !s->gen_now) {
    // the chunks that lag behind the factorisation run on beside the launch above (disjoint rows);
    // everything on the second stream joins before the slice step reads nu
    GP_HIP(hipEventRecord(s->ev_join, s->st2));
    GP_HIP(hipStreamWaitEvent(s->st, s->ev_join, 0));
    if (s->fs_lowrank && s->theta_inside && s->fs_rank > 0) {
      // behind the join: the slice kernels do not wait for it, the f* step does
      GP_TRY(fstar_lowrank_prefactor(s, s->st2));
      GP_HIP(hipEventRecord(s->ev_pref, s->st2));
      s->pref_on_st2 = 1;
    }
  }
  if (s->span_open) {
    GP_HIP(hipEventRecord(s->ev_span[s->span_open - 1][1], s->st));
    s->span_open = 0;
  }
  GP_TRY(mark(s, PH_ESSF));
  EssFArgs a{};
  a.f = s->d_f; a.nu = s->d_nu; a.ld = s->Np_pad; a.mu = nullptr; a.ld_mu = 0;
  a.y = s->d_y; a.ldy = s->Np_pad; a.x = s->d_theta; a.x_stride = s->Np;
  a.beta = s->d_beta; a.bthr = s->d_bthr; a.unif = nullptr;
  a.Np = s->Np; a.U_pad = s->U_pad; a.C = s->C; a.mloc = s->mloc; a.ML = s->ML; a.m = s->m;
  a.j0 = s->j0; a.R = s->cfg.max_ess_rounds; a.seed = s->seed; a.sweep = s->sweep;
  a.rounds = nullptr; a.ll = s->d_llu; a.counters = s->d_counters;
  // few units of many rows: eight waves a unit.  Not under GPIRT_FIXED_ORDER_SUM, whose promise -- the same bits for
  // every number of ranks -- would not survive an order of summation that depends on a rank's share of the units
  a.wide8 = s->Np >= 4096 && s->mloc * s->P <= 1536 && !(s->cfg.flags & GPIRT_FIXED_ORDER_SUM);
  s->share_now = s->share_last = a.wide8 ? 0 : s->share_units;
  if (s->share_now > 0) {
    GP_HIP(hipEventRecord(s->ev_share[0], s->st));
    GP_HIP(hipStreamWaitEvent(s->st_lo, s->ev_share[0], 0));
    a.u0 = 0; a.ucount = s->U_pad - s->share_now;
    GP_TRY(launch_ess_f(s->st, a));
    a.u0 = s->U_pad - s->share_now; a.ucount = s->share_now;
    GP_TRY(launch_ess_f(s->st_lo, a));
    GP_HIP(hipEventRecord(s->ev_share[1], s->st_lo));
    return GPIRT_OK;
  }
  GP_TRY(launch_ess_f(s->st, a));
  return GPIRT_OK;
}

static int phase_param(gpirt_sampler *s) {
  Range range("gpirt:param (beta + thresholds slice update)");
  GP_TRY(mark(s, PH_PARAM));
  EssParamArgs a{};
  a.f = s->d_f; a.ld = s->Np_pad; a.y = s->d_y; a.ldy = s->Np_pad; a.x = s->d_theta;
  a.x_stride = s->Np; a.ebeta = s->d_ebeta; a.delta = s->d_delta; a.beta = s->d_beta;
  a.bthr = s->d_bthr; a.betaS = s->d_betaS; a.bthrS = s->d_bthrS; a.bpm = s->d_bpm; a.bps = s->d_bps;
  a.fixed_thr = s->fixed_thr; a.Np = s->Np; a.U_pad = s->U_pad; a.C = s->C; a.mloc = s->mloc;
  a.ML = s->ML; a.m = s->m; a.j0 = s->j0; a.R = s->cfg.max_ess_rounds; a.seed = s->seed;
  a.sweep = s->sweep; a.ll = s->d_llu; a.counters = s->d_counters + 2;
  a.ll_in = s->d_llu;       // left by phase_f for the same f, beta and thresholds
  if (s->share_now > 0) {
    a.u0 = 0; a.ucount = s->U_pad - s->share_now;
    GP_TRY(launch_ess_param(s->st, a));
    a.u0 = s->U_pad - s->share_now; a.ucount = s->share_now;
    GP_TRY(launch_ess_param(s->st_lo, a));
    GP_HIP(hipEventRecord(s->ev_share[2], s->st_lo));   // joined behind the f* step
    return GPIRT_OK;
  }
  GP_TRY(launch_ess_param(s->st, a));
  return GPIRT_OK;
}

// The part of the low-rank f* step that depends on the inputs only (see setup_fstar_lowrank):
// Phi = P B with sqrt(s2) I appended, its Gram matrix M = s2 I + Phi^T Phi in 128-deep slabs, the
// factor of M and X = M^-1 B^T Lag.  Enqueued on q.
static int fstar_lowrank_prefactor(gpirt_sampler *s, hipStream_t q) {
  const int64_t ldp = s->Np_pad + TILE;
  // Phi = P B, with sqrt(s2) I appended below so that its Gram matrix carries the s2 I
  GemmArgs a{};
  a.A = s->d_W; a.lda = s->Np_pad; a.strideA = (int64_t)s->Np_pad * s->G_pad;
  a.B = s->d_B; a.ldb = s->fs_nodes; a.strideB = 0;
  a.C = s->d_feat; a.ldc = ldp; a.strideC = ldp * TILE;
  a.M = s->Np_pad; a.N = TILE; a.K = s->fs_nodes; a.batch = s->P; a.alpha = 1.0; a.beta = 0.0;
  GP_TRY(launch_gemm(q, false, false, a));
  GP_TRY(launch_scaled_identity(q, s->d_feat, ldp, s->Np_pad, s->P, ldp * TILE, sqrt(s->jitter_now)));
  const int nslab = (int)(ldp / TILE);
  for (int p = 0; p < s->P; ++p) {     // M = Phi^T Phi, K in 128-deep slabs side by side
    GemmArgs m{};
    m.A = s->d_feat + ldp * TILE * p; m.lda = ldp; m.strideA = TILE;
    m.B = m.A; m.ldb = ldp; m.strideB = TILE;
    m.C = s->d_gparts; m.ldc = TILE; m.strideC = TILE * TILE;
    m.M = TILE; m.N = TILE; m.K = TILE; m.batch = nslab; m.alpha = 1.0; m.beta = 0.0;
    GP_TRY(launch_gemm(q, true, false, m));
    GP_TRY(launch_sum_slabs(q, s->d_gparts, nslab, TILE * TILE, TILE, TILE, TILE,
                            s->d_Mm + (int64_t)TILE * TILE * p, TILE));
    GP_HIP(hipMemcpyAsync(s->d_X + (int64_t)TILE * s->G_pad * p, s->d_BL,
                          sizeof(double) * (size_t)TILE * s->G_pad, hipMemcpyDeviceToDevice, q));
  }
  GP_TRY(launch_potrf(q, s->d_Mm, TILE, s->P, s->d_minv, s->d_info + 2, s->d_fwork));
  GP_TRY(launch_potrs(q, s->d_Mm, s->d_minv, TILE, s->P, s->d_X, TILE, (int64_t)TILE * s->G_pad,
                      s->G_pad, s->d_xtmp));
  return GPIRT_OK;
}

// the triangular solves with the n x n factor need the whole inverted super-blocks; the default
// f* route never solves with it, so they are built only when a sweep gets here
static int ensure_super_inverses(gpirt_sampler *s) {
  if (s->super_inv_ready) return GPIRT_OK;
  s->super_inv_ready = 1;
  return launch_super_inverses(s->st, s->d_S, s->Np_pad, s->P, s->d_inv, s->d_fwork, true);
}

static int phase_fstar_body(gpirt_sampler *s);
static int phase_fstar(gpirt_sampler *s) {
  Range range("gpirt:fstar");
  GP_TRY(mark(s, PH_FSTAR));
  // with the slice steps in two shares: f of the second share is awaited here (the step reads every unit's f and
  // parks its partial products in nu), its parameters behind the step, which does not read them
  if (s->share_now > 0) GP_HIP(hipStreamWaitEvent(s->st, s->ev_share[1], 0));
  const int rc = phase_fstar_body(s);
  if (s->share_now > 0) { GP_HIP(hipStreamWaitEvent(s->st, s->ev_share[2], 0)); s->share_now = 0; }
  return rc;
}

static int phase_fstar_body(gpirt_sampler *s) {
  // f*^T = F^T S^-1 K(x, xs)   (gp posterior mean, main.m:57-58).  The solve goes to whichever
  // side has fewer columns: W = S^-1 K(x,xs) (G columns, the same on every rank) or
  // alpha = S^-1 F (this rank's units: fewer than G once the items are sharded, and then no
  // rank repeats another's solve).
  if (s->fs_rank > 0) {
    // interpolated cross-covariance: solve for the R node columns only,
    //   Y = F^T S^-1 K(x, c)   [units x R],   f*^T = Y Lag   [units x G]
    int R = s->fs_rank;
    const bool lowrank = s->fs_lowrank && s->theta_inside;
    const int64_t ldp = s->Np_pad + TILE;
    if (lowrank) {
      // f*^T = (F^T Phi) (s2 I + Phi^T Phi)^-1 (B^T Lag); the factors that depend on the inputs
      // only are ready when they were enqueued on the second stream beside the slice kernels
      if (s->pref_on_st2) GP_HIP(hipStreamWaitEvent(s->st, s->ev_pref, 0));
      else GP_TRY(fstar_lowrank_prefactor(s, s->st));
      R = TILE;
    } else {
      GP_TRY(launch_kstar_nodes(s->st, s->d_theta, s->Np, s->Np, s->Np_pad, s->P, s->cfg.irf_ls,
                                s->cfg.irf_sf, s->d_nodes, s->fs_nodes, R, s->d_W,
                                (int64_t)s->Np_pad * s->G_pad));
      GP_TRY(ensure_super_inverses(s));
      GP_TRY(launch_potrs(s->st, s->d_S, s->d_inv, s->Np_pad, s->P, s->d_W, s->Np_pad,
                          (int64_t)s->Np_pad * s->G_pad, R, s->d_xtmp));
    }
    double *Y = s->d_nu;          // [U_pad, R], free between the f step and the next sweep
    GemmArgs g{};
    g.A = s->d_f; g.lda = s->Np_pad; g.strideA = (int64_t)s->Np_pad * s->ML;
    g.B = s->d_W; g.ldb = s->Np_pad; g.strideB = (int64_t)s->Np_pad * s->G_pad;
    if (lowrank) { g.B = s->d_feat; g.ldb = ldp; g.strideB = ldp * TILE; }
    g.C = Y; g.ldc = s->U_pad; g.strideC = s->ML;
    g.M = s->ML; g.N = R; g.K = s->Np_pad; g.batch = s->P; g.alpha = 1.0; g.beta = 0.0;
    const int64_t tiles = (int64_t)(s->ML / TILE) * (R / TILE);
    int splits = 1;
    if (s->P == 1 && tiles < 512 && s->Np_pad >= 2048) {
      const int slabs = s->Np_pad / 16;
      const int64_t room = ((int64_t)s->Np_pad * s->U_pad) / ((int64_t)s->ML * R);
      // With a rank's share of the items there are only a few output tiles, and eight slabs of K leave each
      // workgroup more than a thousand steps deep on a mostly empty chip (n = 10 000, 512 items: 32 workgroups,
      // 176 us); as many slabs as there are 128-deep blocks of K fill it (316 workgroups, 25 us + 11 us of summing).
      // Under GPIRT_FIXED_ORDER_SUM the slabs stay what every number of ranks would choose.
      const int most = (s->cfg.flags & GPIRT_FIXED_ORDER_SUM) ? 16 : 128;
      for (int q = 2; q <= most && q <= room && tiles * q <= 1024 && s->Np_pad / q >= TILE; ++q)
        if (slabs % q == 0) splits = q;
    }
    if (splits > 1) {   // few output tiles, long K: partial products parked in Z, then summed
      const int kc = s->Np_pad / splits;
      g.K = kc; g.batch = splits; g.strideA = kc; g.strideB = kc;
      g.C = s->d_Z; g.ldc = s->ML; g.strideC = (int64_t)s->ML * R;
      GP_TRY(launch_gemm(s->st, true, false, g));
      GP_TRY(launch_sum_slabs(s->st, s->d_Z, splits, g.strideC, s->ML, R, s->ML, Y, s->U_pad));
    } else {
      GP_TRY(launch_gemm(s->st, true, false, g));
    }
    GemmArgs h{};
    h.A = Y; h.lda = s->U_pad; h.strideA = s->ML;
    h.B = s->d_lag; h.ldb = R; h.strideB = 0;
    if (lowrank) { h.B = s->d_X; h.ldb = TILE; h.strideB = (int64_t)TILE * s->G_pad; }
    h.C = s->d_Gt; h.ldc = s->U_pad; h.strideC = s->ML;
    h.M = s->ML; h.N = s->G_pad; h.K = R; h.batch = s->P; h.alpha = 1.0; h.beta = 0.0;
    GP_TRY(launch_gemm(s->st, false, false, h));
    return GPIRT_OK;
  }
  GP_TRY(launch_kstar_t(s->st, s->d_theta, s->Np, s->Np, s->Np_pad, s->P, s->cfg.irf_ls,
                        s->cfg.irf_sf, s->G, s->G_pad, s->cfg.grid_lo, s->cfg.grid_step, s->d_W));
  const double *lhs = s->d_f;
  if (s->ML < s->G_pad) {
    // nu is free between the f step and the next sweep: alpha is built there
    GP_HIP(hipMemcpyAsync(s->d_nu, s->d_f, sizeof(double) * (size_t)s->Np_pad * s->U_pad,
                          hipMemcpyDeviceToDevice, s->st));
    GP_TRY(ensure_super_inverses(s));
    GP_TRY(launch_potrs(s->st, s->d_S, s->d_inv, s->Np_pad, s->P, s->d_nu, s->Np_pad,
                        (int64_t)s->Np_pad * s->ML, s->ML, s->d_xtmp));
    lhs = s->d_nu;
  } else {
    GP_TRY(ensure_super_inverses(s));
    GP_TRY(launch_potrs(s->st, s->d_S, s->d_inv, s->Np_pad, s->P, s->d_W, s->Np_pad,
                        (int64_t)s->Np_pad * s->G_pad, s->G_pad, s->d_xtmp));
  }
  GemmArgs g{};
  g.A = lhs; g.lda = s->Np_pad; g.strideA = (int64_t)s->Np_pad * s->ML;
  g.B = s->d_W; g.ldb = s->Np_pad; g.strideB = (int64_t)s->Np_pad * s->G_pad;
  g.C = s->d_Gt; g.ldc = s->U_pad; g.strideC = s->ML;
  g.M = s->ML; g.N = s->G_pad; g.K = s->Np_pad; g.batch = s->P; g.alpha = 1.0; g.beta = 0.0;
  // Few output tiles and a long K: split K over the GEMM's batch index (one group only) into
  // partial products parked in Z (free until the next sweep's draws), summed afterwards.
  const int64_t tiles = (int64_t)(s->ML / TILE) * (s->G_pad / TILE);
  int splits = 1;
  if (s->P == 1 && tiles < 512 && s->Np_pad >= 2048) {
    const int slabs = s->Np_pad / 16;
    const int64_t room = ((int64_t)s->Np_pad * s->U_pad) / ((int64_t)s->ML * s->G_pad);
    for (int q = 2; q <= 16 && q <= room && tiles * q <= 1024; ++q)
      if (slabs % q == 0) splits = q;
  }
  if (splits > 1) {
    const int kc = s->Np_pad / splits;
    g.K = kc; g.batch = splits; g.strideA = kc; g.strideB = kc;
    g.C = s->d_Z; g.ldc = s->ML; g.strideC = (int64_t)s->ML * s->G_pad;
    GP_TRY(launch_gemm(s->st, true, false, g));
    GP_TRY(launch_sum_slabs(s->st, s->d_Z, splits, g.strideC, s->ML, s->G_pad, s->ML, s->d_Gt, s->U_pad));
    return GPIRT_OK;
  }
  GP_TRY(launch_gemm(s->st, true, false, g));
  return GPIRT_OK;
}

// The 128 x 128 factorisation of the low-rank f* step reports through d_info[2]; h_info2 is the
// host copy taken behind this sweep's f* step.  A failure ends the sweep that consumed the f*.
static int fstar_status(gpirt_sampler *s) {
  if (!s->h_info2) return GPIRT_OK;
  set_error("low-rank f* step: s2 I + Phi^T Phi is not positive definite (pivot %d)", s->h_info2);
  s->h_info2 = 0;
  (void)hipMemsetAsync(s->d_info + 2, 0, sizeof(int), s->st);
  return GPIRT_ERR_NOT_PD;
}

static ThetaArgs theta_args(gpirt_sampler *s) {
  ThetaArgs a{};
  a.theta = s->d_theta; a.xc = s->d_xc; a.nu = s->d_tnu; a.prop = s->d_prop; a.hh = s->d_hh;
  a.phi = s->d_phi; a.pmin = s->d_pmin; a.pmax = s->d_pmax; a.done = s->d_done; a.buf = s->d_buf;
  a.Lt = s->d_Lt; a.tpm = s->d_tpm; a.tps = s->d_tps; a.Gt = s->d_Gt; a.betaS = s->d_betaS;
  a.bthrS = s->d_bthrS; a.yT = s->d_yT; a.yTl = s->d_yTl; a.n = s->n; a.H = s->H; a.C = s->C; a.mloc = s->mloc;
  a.ML = s->ML; a.U_pad = s->U_pad; a.G = s->G; a.constant_irf = s->cfg.constant_irf;
  a.R = s->cfg.max_ess_rounds; a.grid_lo = s->cfg.grid_lo; a.grid_step = s->cfg.grid_step;
  a.index_truncate = (s->cfg.flags & GPIRT_INDEX_TRUNCATE) ? 1 : 0;
  a.seed = s->seed; a.sweep = s->sweep; a.remaining = s->d_remaining;
  a.counters = s->d_counters + 4;
  a.ll_seen = (s->cfg.flags & GPIRT_FIXED_ORDER_SUM) ? s->d_counters + 6 : nullptr;
  a.tab = s->d_tab;
  a.part = s->d_part; a.gidx = s->d_gidx; a.order = s->d_order; a.bstart = s->d_bstart; a.bcur = s->d_bcur;
  a.ghist = s->d_ghist;
  a.chunk = s->th_chunk; a.nchunks = s->th_nchunks;
  a.slot0 = s->th_slot0; a.nslots = s->th_nslots;
  a.os2 = s->cfg.theta_os > 0 ? s->cfg.theta_os * s->cfg.theta_os : 1.0;
  return a;
}

// sum the per-respondent partial log-likelihoods over the ranks that share the chain
static int exchange(gpirt_sampler *s, double *buf = nullptr, int64_t count = 0) {
  if (s->cfg.world <= 1) return GPIRT_OK;
  if (!(s->cfg.flags & GPIRT_ALLREDUCE_ON_STREAM)) GP_HIP(hipStreamSynchronize(s->st));
  if (!buf) { buf = s->d_buf; count = s->n; }
  if (s->cfg.allreduce(buf, count, s->cfg.allreduce_user)) {
    set_error("allreduce hook failed");
    return GPIRT_ERR_COMM;
  }
  return GPIRT_OK;
}

// Lock-step rounds of THETA_SPEC proposals each.  The decision kernel of a round leaves the
// number of respondents still active in remaining[round]; the host looks at that history only
// after every second round (a look is a stream drain, one copy into pinned memory and a launch
// gap), except with sharded items and a blocking hook, where every round passes through the
// host for the exchange anyway.
static int phase_theta(gpirt_sampler *s) {
  Range range("gpirt:theta (respondent slice update)");
  GP_TRY(mark(s, PH_THETA));
  ThetaArgs a = theta_args(s);
  const int R = a.R;
  // a hook that is not stream-ordered passes through the host every round anyway
  const bool sharded = s->cfg.world > 1 && !(s->cfg.flags & GPIRT_ALLREDUCE_ON_STREAM);
  GP_HIP(hipMemsetAsync(s->d_remaining, 0, sizeof(int) * (size_t)(R + 4), s->st));
  if (a.part) GP_HIP(hipMemsetAsync(s->d_ghist, 0, sizeof(int32_t) * (size_t)s->G, s->st));
  GP_TRY(launch_theta_prepare(s->st, a));
  if (a.tab) GP_TRY(launch_theta_table(s->st, a));
  // a workgroup per respondent pays off once a respondent has enough items to keep four waves
  // busy, and always when few respondents are still active (latency of one respondent's loop)
  const bool many_items = s->mloc * s->H >= 1024;
  // table variant, H = 1: while a good share of the respondents is active, evaluate them
  // grouped by grid index (each table row is then read once per round)
  int active = s->n;
  // GPIRT_FIXED_ORDER_SUM with several ranks: every evaluation is the bucketed one, the ranks exchange
  // the block sums themselves (each rank's blocks in its own slots, zeros in the others': the hook's
  // sum returns every block exactly) and the decision kernel adds all blocks in item order
  // (with one rank the flag alone forces the same evaluation and the same order of adding: nothing to exchange)
  const bool fixed_order = (s->cfg.flags & GPIRT_FIXED_ORDER_SUM) != 0;
  const bool fixed = fixed_order && s->cfg.world > 1;
  const int64_t slot = (int64_t)THETA_SPEC * s->n;
  auto eval = [&](int initial, bool wide) -> int {
    const bool bucketed = a.part && (fixed_order || active * 8 >= s->n);
    a.from_parts = bucketed && (s->cfg.world <= 1 || fixed);   // read by the decision kernel that follows
    if (fixed) {
      if (a.slot0 > 0) GP_HIP(hipMemsetAsync(s->d_part, 0, sizeof(double) * (size_t)(slot * a.slot0), s->st));
      const int after = a.slot0 + a.nchunks;
      if (after < a.nslots)
        GP_HIP(hipMemsetAsync(s->d_part + slot * after, 0, sizeof(double) * (size_t)(slot * (a.nslots - after)), s->st));
    }
    if (bucketed) return launch_theta_eval_bucketed(s->st, a, initial);
    return launch_theta_eval(s->st, a, initial, wide);
  };
  auto share = [&](int initial) -> int {
    if (fixed) return exchange(s, s->d_part, slot * a.nslots);
    return initial ? exchange(s) : exchange(s, s->d_buf, slot);
  };
  GP_TRY(eval(1, many_items));
  GP_TRY(share(1));
  GP_TRY(launch_theta_decide(s->st, a, 0, 0));
  const int max_rounds = (R + THETA_SPEC - 1) / THETA_SPEC + 1;
  std::vector<int> hist((size_t)R + 4, 0);
  int r = 1;
  bool finished = false;
  while (r <= max_rounds && !finished) {
    // a round nobody needs any more costs three kernels that return at once, less than a look (~35 us of idle
    // GPU): the first four rounds go out together -- hardly a sweep needs fewer -- then three at a time
    const int last = sharded ? r : min(max_rounds, r < 2 ? 4 : r + 2);
    const bool wide = many_items || active * 4 <= s->n;
    for (int q = r; q <= last; ++q) {
      GP_TRY(eval(0, wide));
      GP_TRY(share(0));
      GP_TRY(launch_theta_decide(s->st, a, 1 + (q - 1) * THETA_SPEC, q));
    }
    // one copy into pinned memory brings the status words and the whole history of active counts
    GP_HIP(hipMemcpyAsync(s->h_look, s->d_info, sizeof(int) * (size_t)(4 + R + 4), hipMemcpyDeviceToHost, s->st));
    GP_HIP(hipStreamSynchronize(s->st));
    for (int q = r; q <= last; ++q) hist[q] = s->h_look[4 + q];
    hist[(size_t)R + 3] = s->h_look[4 + R + 3];
    if (s->fs_lowrank) s->h_info2 = s->h_look[2];
    GP_TRY(fstar_status(s));
    for (int q = r; q <= last; ++q) {
      active = hist[q];
      if (!active) { finished = true; break; }    // rounds after q (if any) found nobody active
    }
    r = last + 1;
  }
  // what one proposal per lock-step round would have counted: the evaluation of the current
  // state plus as many rounds as the slowest respondent needed proposals
  s->theta_rounds += 1 + hist[(size_t)R + 3];
  return GPIRT_OK;
}

// static respondent step (utils/sample_score.m:1-20): table, lp = T * onehot, draw
static int phase_theta_grid(gpirt_sampler *s) {
  Range range("gpirt:theta (grid inverse-CDF)");
  GP_TRY(mark(s, PH_THETA));
  ThetaArgs a = theta_args(s);
  GP_TRY(launch_grid_table(s->st, a, s->G_pad, s->d_T));
  GemmArgs g{};
  g.A = s->d_T; g.lda = s->G_pad;
  g.B = s->d_onehot; g.ldb = (int64_t)s->C * s->ML;
  g.C = s->d_lp; g.ldc = s->G_pad;
  g.M = s->G_pad; g.N = s->Np_pad; g.K = s->C * s->ML; g.batch = 1; g.alpha = 1.0; g.beta = 0.0;
  GP_TRY(launch_gemm(s->st, false, false, g));
  GP_TRY(exchange(s, s->d_lp, (int64_t)s->G_pad * s->n));
  s->theta_rounds++;
  GP_TRY(launch_grid_pick(s->st, a, s->d_lp, s->G_pad));
  return GPIRT_OK;
}

static int phase_snap(gpirt_sampler *s) {
  Range range("gpirt:snap");
  GP_TRY(mark(s, PH_SNAP));
  GP_TRY(launch_snap(s->st, s->d_Gt, s->U_pad, s->d_theta, s->Np, s->Np, s->P, s->mloc, s->ML, s->G,
                     s->cfg.grid_lo, s->cfg.grid_step, (s->cfg.flags & GPIRT_INDEX_TRUNCATE) ? 1 : 0, s->d_f,
                     s->Np_pad));
  return GPIRT_OK;
}

extern "C" {

int gpirt_sampler_sweep(gpirt_sampler *s, int32_t sweeps) {
  if (!s) { set_error("sampler is NULL"); return GPIRT_ERR_BAD_ARG; }
  GP_HIP(hipSetDevice(s->device));
  choose_slice_shares(s);
  for (int it = 0; it < sweeps; ++it) {
    Range range("gpirt:sweep");
    GP_TRY(phase_factor(s));   // main.m:67 + the chol of sample_fs.m:2
    GP_TRY(phase_f(s));        // main.m:50-52
    GP_TRY(phase_param(s));
    GP_TRY(phase_fstar(s));    // main.m:55-60
    GP_TRY(s->cfg.theta_mode == GPIRT_THETA_GRID ? phase_theta_grid(s) : phase_theta(s));   // main.m:63
    GP_TRY(phase_snap(s));     // main.m:69-73
    GP_TRY(mark(s, -1));
    s->sweep++;
    if (s->ev_used > 4096 || s->upd_used > 4096) {
      GP_HIP(hipStreamSynchronize(s->st));
      GP_TRY(collect_times(s));
    }
  }
  // the last sweep's f* status would otherwise be seen by the next call only (grid-mode
  // respondent steps never look at it inside the sweep)
  if (s->fs_lowrank && sweeps > 0)
    GP_HIP(hipMemcpyAsync(s->h_look, s->d_info, 4 * sizeof(int), hipMemcpyDeviceToHost, s->st));
  GP_HIP(hipStreamSynchronize(s->st));
  GP_TRY(collect_times(s));
  if (s->fs_lowrank && sweeps > 0) s->h_info2 = s->h_look[2];
  GP_TRY(fstar_status(s));
  return GPIRT_OK;
}

int gpirt_sampler_get_state(gpirt_sampler *s, double *theta, double *f, double *fstar, double *beta,
                            double *threshold) {
  if (!s) { set_error("sampler is NULL"); return GPIRT_ERR_BAD_ARG; }
  GP_HIP(hipSetDevice(s->device));
  const int n = s->n, H = s->H, C = s->C, P = s->P, mloc = s->mloc, ML = s->ML, U_pad = s->U_pad,
            Np_pad = s->Np_pad, G = s->G;
  GP_HIP(hipStreamSynchronize(s->st));
  if (theta) GP_HIP(hipMemcpy(theta, s->d_theta, sizeof(double) * n * H, hipMemcpyDeviceToHost));
  if (f) {
    if (!s->cfg.constant_irf) {
      for (int h = 0; h < H; ++h)
        GP_HIP(hipMemcpy2D(f + (size_t)n * mloc * h, sizeof(double) * n,
                           s->d_f + (size_t)Np_pad * ML * h, sizeof(double) * Np_pad,
                           sizeof(double) * n, mloc, hipMemcpyDeviceToHost));
    } else {
      std::vector<double> tmp((size_t)Np_pad * mloc);
      GP_HIP(hipMemcpy(tmp.data(), s->d_f, sizeof(double) * tmp.size(), hipMemcpyDeviceToHost));
      for (int h = 0; h < H; ++h)
        for (int jl = 0; jl < mloc; ++jl)
          memcpy(f + (size_t)n * (jl + (size_t)mloc * h), &tmp[(size_t)n * h + (size_t)Np_pad * jl],
                 sizeof(double) * n);
    }
  }
  if (fstar) {
    DevBuf tmp;
    GP_TRY(tmp.alloc(sizeof(double) * (size_t)G * mloc * P, false));
    GP_TRY(launch_export_fstar(s->st, s->d_Gt, U_pad, G, P, mloc, ML, tmp.as<double>()));
    GP_HIP(hipStreamSynchronize(s->st));
    GP_HIP(hipMemcpy(fstar, tmp.p, sizeof(double) * (size_t)G * mloc * P, hipMemcpyDeviceToHost));
  }
  if (beta) {
    for (int p = 0; p < P; ++p)
      GP_HIP(hipMemcpy(beta + 3 * (size_t)mloc * p, s->d_beta + 3 * (size_t)ML * p,
                       sizeof(double) * 3 * mloc, hipMemcpyDeviceToHost));
  }
  if (threshold) {
    std::vector<double> hb((size_t)(C + 1) * U_pad);
    GP_HIP(hipMemcpy(hb.data(), s->d_bthr, sizeof(double) * hb.size(), hipMemcpyDeviceToHost));
    for (int p = 0; p < P; ++p)
      for (int c = 0; c <= C; ++c)
        for (int jl = 0; jl < mloc; ++jl)
          threshold[jl + (size_t)mloc * (c + (size_t)(C + 1) * p)] =
              hb[c + (size_t)(C + 1) * (jl + (size_t)ML * p)];
  }
  return GPIRT_OK;
}

int gpirt_sampler_phase_times(gpirt_sampler *s, double *ms, int64_t *count) {
  if (!s) { set_error("sampler is NULL"); return GPIRT_ERR_BAD_ARG; }
  for (int k = 0; k < PH_COUNT; ++k) {
    if (ms) ms[k] = s->phase_ms[k];
    if (count) count[k] = s->phase_cnt[k];
  }
  return GPIRT_OK;
}

int gpirt_sampler_update_times(gpirt_sampler *s, double *ms, int64_t *launches, double *flop) {
  if (!s) { set_error("sampler is NULL"); return GPIRT_ERR_BAD_ARG; }
  if (ms) *ms = s->upd_ms;
  if (launches) *launches = s->upd_cnt;
  if (flop) *flop = s->upd_flop;
  return GPIRT_OK;
}

int gpirt_sampler_diagnostics(gpirt_sampler *s, gpirt_outputs *o) {
  if (!s || !o) { set_error("NULL argument"); return GPIRT_ERR_BAD_ARG; }
  GP_HIP(hipSetDevice(s->device));
  unsigned long long c[6];
  GP_HIP(hipStreamSynchronize(s->st));
  GP_HIP(hipMemcpy(c, s->d_counters, sizeof c, hipMemcpyDeviceToHost));
  o->ess_f_rounds = (int64_t)c[0];
  o->ess_param_rounds = (int64_t)c[2];
  o->ess_theta_rounds = s->theta_rounds;
  o->ess_capped = (int64_t)(c[1] + c[3] + c[5]);
  o->jitter_retries = s->jitter_retries;
  o->nu_rows_beside_factor = s->gen_now ? 0 : s->nu_rows_done;
  o->factor_lookahead = s->gen_now ? 0 : s->la_now;
  o->nu_tail_panels = s->gen_now ? 0 : s->nu_tail_done;
  o->slice_second_share = s->share_last;
  return GPIRT_OK;
}

int gpirt_sampler_ll_checksum(gpirt_sampler *s, uint64_t *out) {
  if (!s || !out) { set_error("NULL argument"); return GPIRT_ERR_BAD_ARG; }
  if (!(s->cfg.flags & GPIRT_FIXED_ORDER_SUM)) {
    set_error("gpirt_sampler_ll_checksum: the sampler was not created with GPIRT_FIXED_ORDER_SUM");
    return GPIRT_ERR_BAD_ARG;
  }
  GP_HIP(hipSetDevice(s->device));
  GP_HIP(hipStreamSynchronize(s->st));
  unsigned long long v = 0;
  GP_HIP(hipMemcpy(&v, s->d_counters + 6, sizeof v, hipMemcpyDeviceToHost));
  *out = (uint64_t)v;
  return GPIRT_OK;
}

// ---- checkpoint / resume ----------------------------------------------------------------
namespace {
struct CkptHeader {
  uint64_t magic;
  int32_t n, m, H, C, P, mloc, j0, G;
  uint64_t seed;
  uint32_t sweep;
  int32_t jitter_retries;
  double jitter_now;
  int64_t theta_rounds;
  unsigned long long counters[6];
};
constexpr uint64_t kCkptMagic = 0x3154504b43525047ull;   // "GPRCKPT1"

struct CkptPart { void *dev; size_t bytes; };
void ckpt_parts(gpirt_sampler *s, std::vector<CkptPart> &parts) {
  const size_t U = s->U_pad, C = s->C;
  parts = {
      {s->d_theta, sizeof(double) * (size_t)s->n * s->H},
      {s->d_f, sizeof(double) * (size_t)s->Np_pad * U},
      {s->d_Gt, sizeof(double) * U * s->G_pad},
      {s->d_ebeta, sizeof(double) * 3 * U},
      {s->d_delta, sizeof(double) * (C - 1) * U},
      {s->d_beta, sizeof(double) * 3 * U},
      {s->d_bthr, sizeof(double) * (C + 1) * U},
      {s->d_betaS, sizeof(double) * 3 * U},
      {s->d_bthrS, sizeof(double) * (C + 1) * U},
  };
}
}  // namespace

int64_t gpirt_sampler_checkpoint_bytes(gpirt_sampler *s) {
  if (!s) return 0;
  size_t total = sizeof(CkptHeader);
  std::vector<CkptPart> parts;
  ckpt_parts(s, parts);
  for (const CkptPart &p : parts) total += p.bytes;
  return (int64_t)total;
}

int gpirt_sampler_checkpoint(gpirt_sampler *s, void *blob, int64_t bytes) {
  if (!s || !blob) { set_error("sampler/blob is NULL"); return GPIRT_ERR_BAD_ARG; }
  if (bytes < gpirt_sampler_checkpoint_bytes(s)) {
    set_error("checkpoint buffer too small: %lld < %lld", (long long)bytes,
              (long long)gpirt_sampler_checkpoint_bytes(s));
    return GPIRT_ERR_BAD_ARG;
  }
  GP_HIP(hipSetDevice(s->device));
  GP_HIP(hipStreamSynchronize(s->st));
  CkptHeader h{};
  h.magic = kCkptMagic;
  h.n = s->n; h.m = s->m; h.H = s->H; h.C = s->C; h.P = s->P; h.mloc = s->mloc; h.j0 = s->j0; h.G = s->G;
  h.seed = s->seed; h.sweep = s->sweep; h.jitter_retries = s->jitter_retries;
  h.jitter_now = s->jitter_now; h.theta_rounds = s->theta_rounds;
  GP_HIP(hipMemcpy(h.counters, s->d_counters, sizeof h.counters, hipMemcpyDeviceToHost));
  char *out = static_cast<char *>(blob);
  memcpy(out, &h, sizeof h);
  out += sizeof h;
  std::vector<CkptPart> parts;
  ckpt_parts(s, parts);
  for (const CkptPart &p : parts) {
    GP_HIP(hipMemcpy(out, p.dev, p.bytes, hipMemcpyDeviceToHost));
    out += p.bytes;
  }
  return GPIRT_OK;
}

int gpirt_sampler_restore(gpirt_sampler *s, const void *blob, int64_t bytes) {
  if (!s || !blob) { set_error("sampler/blob is NULL"); return GPIRT_ERR_BAD_ARG; }
  if (bytes < gpirt_sampler_checkpoint_bytes(s)) {
    set_error("checkpoint blob too small for this sampler");
    return GPIRT_ERR_BAD_ARG;
  }
  CkptHeader h;
  memcpy(&h, blob, sizeof h);
  if (h.magic != kCkptMagic || h.n != s->n || h.m != s->m || h.H != s->H || h.C != s->C || h.P != s->P ||
      h.mloc != s->mloc || h.j0 != s->j0 || h.G != s->G || h.seed != s->seed) {
    set_error("checkpoint does not belong to a sampler of this shape, item range and seed");
    return GPIRT_ERR_BAD_ARG;
  }
  GP_HIP(hipSetDevice(s->device));
  GP_HIP(hipStreamSynchronize(s->st));
  const char *in = static_cast<const char *>(blob) + sizeof h;
  std::vector<CkptPart> parts;
  ckpt_parts(s, parts);
  for (const CkptPart &p : parts) {
    GP_HIP(hipMemcpy(p.dev, in, p.bytes, hipMemcpyHostToDevice));
    in += p.bytes;
  }
  GP_HIP(hipMemcpy(s->d_counters, h.counters, sizeof h.counters, hipMemcpyHostToDevice));
  s->sweep = h.sweep; s->jitter_retries = h.jitter_retries; s->jitter_now = h.jitter_now;
  s->theta_rounds = h.theta_rounds;
  return GPIRT_OK;
}

int gpirt_sampler_accumulate(gpirt_sampler *s, int32_t options) {
  const int quadratic = options & GPIRT_SUMMARY_QUADRATIC;
  if (!s) { set_error("sampler is NULL"); return GPIRT_ERR_BAD_ARG; }
  GP_HIP(hipSetDevice(s->device));
  if (!s->d_sum_th) {
    GP_TRY(dalloc(s->d_sum_th, (size_t)s->n * s->H));
    GP_TRY(dalloc(s->d_sum_th2, (size_t)s->n * s->H));
    GP_TRY(dalloc(s->d_sum_p, (size_t)s->U_pad * s->C * s->G));
    GP_TRY(dalloc(s->d_sum_arg, (size_t)s->U_pad * s->G));
  }
  if (!(options & GPIRT_SUMMARY_SKIP_THETA)) {
    GP_TRY(launch_summary_theta(s->st, s->d_theta, (int64_t)s->n * s->H, s->d_sum_th, s->d_sum_th2));
    s->summary_theta_draws++;
  }
  ThetaArgs a = theta_args(s);
  GP_TRY(launch_summary_prob(s->st, a, quadratic != 0, s->d_sum_p, s->d_sum_arg));
  GP_HIP(hipStreamSynchronize(s->st));
  s->summary_draws++;
  return GPIRT_OK;
}

int gpirt_sampler_summary(gpirt_sampler *s, gpirt_summary *out) {
  if (!s || !out) { set_error("sampler/out is NULL"); return GPIRT_ERR_BAD_ARG; }
  if (out->struct_size != (int32_t)sizeof(gpirt_summary)) {
    set_error("gpirt_summary.struct_size %d != %zu (header mismatch)", out->struct_size, sizeof(gpirt_summary));
    return GPIRT_ERR_BAD_ARG;
  }
  GP_HIP(hipSetDevice(s->device));
  int K = s->summary_draws;
  out->draws = K;
  if (K < 1) { set_error("no draw has been accumulated"); return GPIRT_ERR_BAD_ARG; }
  const size_t nH = (size_t)s->n * s->H;
  if (out->theta_mean || out->theta_sd) {
    K = s->summary_theta_draws;          // draws whose theta was accumulated
    if (K < 1) { set_error("no draw's theta has been accumulated"); return GPIRT_ERR_BAD_ARG; }
    std::vector<double> a(nH), b(nH);
    GP_HIP(hipMemcpy(a.data(), s->d_sum_th, sizeof(double) * nH, hipMemcpyDeviceToHost));
    GP_HIP(hipMemcpy(b.data(), s->d_sum_th2, sizeof(double) * nH, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < nH; ++i) {
      const double mean = a[i] / K;
      if (out->theta_mean) out->theta_mean[i] = mean;
      if (out->theta_sd) {
        const double var = K > 1 ? (b[i] - K * mean * mean) / (K - 1) : NAN;   // R's sd(): n-1
        out->theta_sd[i] = K > 1 ? sqrt(var > 0.0 ? var : 0.0) : NAN;
      }
    }
    K = s->summary_draws;
  }
  const int G = s->G, C = s->C, P = s->P, mloc = s->mloc, ML = s->ML, U_pad = s->U_pad;
  if (out->prob_mean) {
    std::vector<double> h((size_t)U_pad * C * G);
    GP_HIP(hipMemcpy(h.data(), s->d_sum_p, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < P; ++p)
        for (int jl = 0; jl < mloc; ++jl)
          for (int g = 0; g < G; ++g)
            out->prob_mean[g + (size_t)G * (jl + (size_t)mloc * (p + (size_t)P * c))] =
                h[(size_t)(jl + ML * p) + (size_t)U_pad * (c + (size_t)C * g)] / K;
  }
  if (out->argmax_mean) {
    std::vector<double> h((size_t)U_pad * G);
    GP_HIP(hipMemcpy(h.data(), s->d_sum_arg, sizeof(double) * h.size(), hipMemcpyDeviceToHost));
    for (int p = 0; p < P; ++p)
      for (int jl = 0; jl < mloc; ++jl)
        for (int g = 0; g < G; ++g)
          out->argmax_mean[g + (size_t)G * (jl + (size_t)mloc * p)] =
              h[(size_t)(jl + ML * p) + (size_t)U_pad * g] / K;
  }
  return GPIRT_OK;
}

void *gpirt_sampler_stream(gpirt_sampler *s) { return s ? (void *)s->st : nullptr; }

void gpirt_sampler_destroy(gpirt_sampler *s) {
  if (s) {
    (void)hipSetDevice(s->device);
    (void)hipStreamSynchronize(s->st);
  }
  sampler_free(s);
}

// end-of-sweep log-likelihood of this rank's items at the current (theta, f, beta, b)
static int sampler_loglik(gpirt_sampler *s, double *out) {
  // g = mu + f is evaluated inside the per-unit slice kernel with zero rounds allowed:
  // R = 0 makes it return the log-likelihood of the current state.
  EssFArgs a{};
  a.f = s->d_f; a.nu = s->d_f; a.ld = s->Np_pad; a.y = s->d_y; a.ldy = s->Np_pad; a.x = s->d_theta;
  a.x_stride = s->Np; a.beta = s->d_beta; a.bthr = s->d_bthr; a.Np = s->Np; a.U_pad = s->U_pad;
  a.C = s->C; a.mloc = s->mloc; a.ML = s->ML; a.m = s->m; a.j0 = s->j0; a.R = 0; a.seed = s->seed;
  a.sweep = s->sweep; a.ll = s->d_llu;
  GP_HIP(hipMemsetAsync(s->d_llu, 0, sizeof(double) * s->U_pad, s->st));
  GP_TRY(launch_ess_f(s->st, a));
  std::vector<double> h(s->U_pad);
  GP_HIP(hipMemcpyAsync(h.data(), s->d_llu, sizeof(double) * s->U_pad, hipMemcpyDeviceToHost, s->st));
  GP_HIP(hipStreamSynchronize(s->st));
  double acc = 0.0;
  for (int p = 0; p < s->P; ++p)
    for (int jl = 0; jl < s->mloc; ++jl) acc += h[jl + (size_t)s->ML * p];
  *out = acc;
  return GPIRT_OK;
}

int gpirt_run(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
              const double *bpm, const double *bps, const double *tpm, const double *tps,
              const double *thresholds, gpirt_outputs *out) {
  return gpirt_run_summary(cfg, y, theta_init, bpm, bps, tpm, tps, thresholds, out, nullptr, 0);
}

int gpirt_run_summary(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
                      const double *bpm, const double *bps, const double *tpm, const double *tps,
                      const double *thresholds, gpirt_outputs *out, gpirt_summary *summary,
                      int32_t options) {
  gpirt_sampler *s = nullptr;
  GP_TRY(gpirt_sampler_create(cfg, y, theta_init, bpm, bps, tpm, tps, thresholds, &s));
  const int total = cfg->burn_iterations + cfg->sample_iterations;
  const int thin = cfg->thin > 0 ? cfg->thin : 1;
  const int K = cfg->sample_iterations / thin;
  int k = 0, rc = GPIRT_OK;
  const size_t nH = (size_t)s->n * s->H, sf = (size_t)s->n * s->mloc * s->H,
               sfs = (size_t)s->G * s->mloc * s->P, sb = 3 * (size_t)s->mloc * s->P,
               stt = (size_t)(s->C + 1) * s->mloc * s->P;
  auto t0 = std::chrono::steady_clock::now();
  for (int it = 0; it < total && rc == GPIRT_OK; ++it) {
    rc = gpirt_sampler_sweep(s, 1);
    if (rc) break;
    if (out && out->loglik) rc = sampler_loglik(s, &out->loglik[it]);
    if (rc) break;
    const int post = it - cfg->burn_iterations;
    const bool keep = post >= 0 && (post + 1) % thin == 0 && k < K;
    if (out && keep) {
      rc = gpirt_sampler_get_state(
          s, out->theta ? out->theta + nH * k : nullptr,
          (out->f && (cfg->flags & GPIRT_RETURN_F)) ? out->f + sf * k : nullptr,
          (out->fstar && (cfg->flags & GPIRT_RETURN_FSTAR)) ? out->fstar + sfs * k : nullptr,
          out->beta ? out->beta + sb * k : nullptr, out->threshold ? out->threshold + stt * k : nullptr);
    }
    if (!rc && summary && keep)   // samples$theta[-1, , ]: the first kept draw's theta may be left out
      rc = gpirt_sampler_accumulate(s, (options & GPIRT_SUMMARY_QUADRATIC) |
                                           ((k == 0 && (options & GPIRT_SUMMARY_DROP_FIRST_THETA)) ? GPIRT_SUMMARY_SKIP_THETA : 0));
    if (keep) ++k;
    if (!rc && cfg->progress && cfg->progress(it + 1, total, cfg->progress_user)) {
      set_error("interrupted by the progress callback at sweep %d", it + 1);
      rc = GPIRT_ERR_INTERRUPT;
    }
  }
  if (!rc && summary) rc = gpirt_sampler_summary(s, summary);
  if (out) {
    int rc2 = gpirt_sampler_diagnostics(s, out);
    if (!rc) rc = rc2;
    out->seconds_total =
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  }
  gpirt_sampler_destroy(s);
  return rc;
}

// ---- parity entry points --------------------------------------------------------------------

int gpirt_k_build(const double *x, int32_t n, double ls, double sf, double jitter, double *K) {
  if (!x || !K || n < 1) { set_error("gpirt_k_build: bad argument"); return GPIRT_ERR_BAD_ARG; }
  GP_TRY(have_device());
  const int np = (int)pad_up(n);
  DevBuf dx, dS;
  GP_TRY(dx.alloc(sizeof(double) * n));
  GP_TRY(dS.alloc(sizeof(double) * (size_t)np * np, false));
  GP_HIP(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
  GP_TRY(launch_k_build(0, dx.as<double>(), 0, n, np, 1, ls, sf, jitter, dS.as<double>(), true));
  GP_TRY(download_padded(K, n, n, dS.as<double>(), np, 0));
  GP_HIP(hipDeviceSynchronize());
  return GPIRT_OK;
}

// upload an n x n matrix into a padded buffer whose pad block is the identity
static int upload_spd_padded(const double *S, int n, int np, double *d) {
  GP_HIP(hipMemset(d, 0, sizeof(double) * (size_t)np * np));
  GP_TRY(upload_padded(S, n, n, d, np, 0));
  if (np > n) {
    std::vector<double> ones(np - n, 1.0);
    GP_HIP(hipMemcpy2D(d + (size_t)n * (np + 1), sizeof(double) * (np + 1), ones.data(),
                       sizeof(double), sizeof(double), np - n, hipMemcpyHostToDevice));
  }
  return GPIRT_OK;
}

int gpirt_potrf(const double *S, int32_t n, double *L) {
  if (!S || !L || n < 1) { set_error("gpirt_potrf: bad argument"); return GPIRT_ERR_BAD_ARG; }
  GP_TRY(have_device());
  const int np = (int)pad_up(n);
  DevBuf dS, dI, dinfo, dwork;
  GP_TRY(dS.alloc(sizeof(double) * (size_t)np * np, false));
  GP_TRY(dI.alloc(sizeof(double) * (size_t)inv_group_elems(np)));
  GP_TRY(dwork.alloc(sizeof(double) * (size_t)potrf_work_elems(np, 1)));
  GP_TRY(dinfo.alloc(sizeof(int)));
  GP_TRY(upload_spd_padded(S, n, np, dS.as<double>()));
  GP_TRY(launch_potrf(0, dS.as<double>(), np, 1, dI.as<double>(), dinfo.as<int>(), dwork.as<double>(), nullptr,
                      POTRF_DEFER_INVERSES | POTRF_SUBSTITUTION));    // as the sweep factors
  int info = 0;
  GP_HIP(hipMemcpy(&info, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
  GP_TRY(download_padded(L, n, n, dS.as<double>(), np, 0));
  GP_HIP(hipDeviceSynchronize());
  for (int j = 1; j < n; ++j)
    for (int i = 0; i < j; ++i) L[i + (size_t)n * j] = 0.0;
  if (info) { set_error("gpirt_potrf: pivot %d is not positive", info); return GPIRT_ERR_NOT_PD; }
  return GPIRT_OK;
}

int gpirt_trmm_lz(const double *L, const double *Z, int32_t n, int32_t k, double *N) {
  if (!L || !Z || !N || n < 1 || k < 1) { set_error("gpirt_trmm_lz: bad argument"); return GPIRT_ERR_BAD_ARG; }
  GP_TRY(have_device());
  const int np = (int)pad_up(n), kp = (int)pad_up(k);
  DevBuf dL, dZ, dN;
  GP_TRY(dL.alloc(sizeof(double) * (size_t)np * np));
  GP_TRY(dZ.alloc(sizeof(double) * (size_t)np * kp));
  GP_TRY(dN.alloc(sizeof(double) * (size_t)np * kp));
  GP_TRY(upload_padded(L, n, n, dL.as<double>(), np, 0));
  GP_TRY(upload_padded(Z, n, k, dZ.as<double>(), np, 0));
  GemmArgs g{};
  g.A = dL.as<double>(); g.lda = np; g.B = dZ.as<double>(); g.ldb = np; g.C = dN.as<double>();
  g.ldc = np; g.M = np; g.N = kp; g.K = np; g.batch = 1; g.alpha = 1.0; g.beta = 0.0; g.k_trim = 1;
  GP_TRY(launch_gemm(0, false, false, g));
  GP_TRY(download_padded(N, n, k, dN.as<double>(), np, 0));
  GP_HIP(hipDeviceSynchronize());
  return GPIRT_OK;
}

// device factor + diagonal-block inverses from a host factor (re-derives the inverses by
// factoring L L^T's diagonal blocks is not possible without S, so S = L L^T is rebuilt)
static int factor_from_host_L(const double *L, int n, int np, DevBuf &dS, DevBuf &dI) {
  DevBuf dL, dinfo, dwork;
  GP_TRY(dL.alloc(sizeof(double) * (size_t)np * np));
  GP_TRY(dS.alloc(sizeof(double) * (size_t)np * np));
  GP_TRY(dI.alloc(sizeof(double) * (size_t)inv_group_elems(np)));
  GP_TRY(dwork.alloc(sizeof(double) * (size_t)potrf_work_elems(np, 1)));
  GP_TRY(dinfo.alloc(sizeof(int)));
  GP_TRY(upload_spd_padded(L, n, np, dL.as<double>()));
  GemmArgs g{};  // S = L L^T
  g.A = dL.as<double>(); g.lda = np; g.B = dL.as<double>(); g.ldb = np; g.C = dS.as<double>();
  g.ldc = np; g.M = np; g.N = np; g.K = np; g.batch = 1; g.alpha = 1.0; g.beta = 0.0;
  GP_TRY(launch_gemm(0, false, true, g));
  GP_TRY(launch_potrf(0, dS.as<double>(), np, 1, dI.as<double>(), dinfo.as<int>(), dwork.as<double>()));
  int info = 0;
  GP_HIP(hipMemcpy(&info, dinfo.p, sizeof(int), hipMemcpyDeviceToHost));
  if (info) { set_error("L L^T is not positive definite (pivot %d)", info); return GPIRT_ERR_NOT_PD; }
  return GPIRT_OK;
}

int gpirt_potrs(const double *L, const double *B, int32_t n, int32_t k, double *X) {
  if (!L || !B || !X || n < 1 || k < 1) { set_error("gpirt_potrs: bad argument"); return GPIRT_ERR_BAD_ARG; }
  GP_TRY(have_device());
  const int np = (int)pad_up(n), kp = (int)pad_up(k);
  DevBuf dS, dI, dB, dX;
  GP_TRY(factor_from_host_L(L, n, np, dS, dI));
  GP_TRY(dB.alloc(sizeof(double) * (size_t)np * kp));
  GP_TRY(dX.alloc(sizeof(double) * (size_t)4 * SUP * kp));
  GP_TRY(upload_padded(B, n, k, dB.as<double>(), np, 0));
  GP_TRY(launch_potrs(0, dS.as<double>(), dI.as<double>(), np, 1, dB.as<double>(), np, 0, kp,
                      dX.as<double>()));
  GP_TRY(download_padded(X, n, k, dB.as<double>(), np, 0));
  GP_HIP(hipDeviceSynchronize());
  return GPIRT_OK;
}

int gpirt_ordinal_loglik(const double *g, const int8_t *y, const double *b, int32_t n, int32_t u,
                         int32_t C, double *ll_unit, double *ll_row) {
  if (!g || !y || !b || n < 1 || u < 1 || C < 2 || C > MAXC) {
    set_error("gpirt_ordinal_loglik: bad argument");
    return GPIRT_ERR_BAD_ARG;
  }
  GP_TRY(have_device());
  DevBuf dg, dy, db, dlu, dlr;
  GP_TRY(dg.alloc(sizeof(double) * (size_t)n * u));
  GP_TRY(dy.alloc((size_t)n * u));
  GP_TRY(db.alloc(sizeof(double) * (size_t)(C + 1) * u));
  GP_TRY(dlu.alloc(sizeof(double) * u));
  GP_TRY(dlr.alloc(sizeof(double) * n));
  GP_HIP(hipMemcpy(dg.p, g, sizeof(double) * (size_t)n * u, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(dy.p, y, (size_t)n * u, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(db.p, b, sizeof(double) * (size_t)(C + 1) * u, hipMemcpyHostToDevice));
  GP_TRY(launch_loglik(0, dg.as<double>(), n, dy.as<int8_t>(), n, db.as<double>(), n, u, C,
                       ll_unit ? dlu.as<double>() : nullptr, ll_row ? dlr.as<double>() : nullptr));
  if (ll_unit) GP_HIP(hipMemcpy(ll_unit, dlu.p, sizeof(double) * u, hipMemcpyDeviceToHost));
  if (ll_row) GP_HIP(hipMemcpy(ll_row, dlr.p, sizeof(double) * n, hipMemcpyDeviceToHost));
  GP_HIP(hipDeviceSynchronize());
  return GPIRT_OK;
}

int gpirt_ess_f(double *f, const double *nu, const double *mu, const int8_t *y, const double *b,
                const double *unif, int32_t n, int32_t u, int32_t C, int32_t max_rounds,
                int32_t *rounds, double *ll) {
  if (!f || !nu || !y || !b || !unif || n < 1 || u < 1 || C < 2 || C > MAXC || max_rounds < 0) {
    set_error("gpirt_ess_f: bad argument");
    return GPIRT_ERR_BAD_ARG;
  }
  GP_TRY(have_device());
  const size_t nu_ = (size_t)n * u;
  DevBuf df, dn, dm, dy, db, du, dr, dl;
  GP_TRY(df.alloc(sizeof(double) * nu_)); GP_TRY(dn.alloc(sizeof(double) * nu_));
  GP_TRY(dm.alloc(sizeof(double) * nu_)); GP_TRY(dy.alloc(nu_));
  GP_TRY(db.alloc(sizeof(double) * (size_t)(C + 1) * u));
  GP_TRY(du.alloc(sizeof(double) * (size_t)(max_rounds + 2) * u));
  GP_TRY(dr.alloc(sizeof(int32_t) * u)); GP_TRY(dl.alloc(sizeof(double) * u));
  GP_HIP(hipMemcpy(df.p, f, sizeof(double) * nu_, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(dn.p, nu, sizeof(double) * nu_, hipMemcpyHostToDevice));
  if (mu) GP_HIP(hipMemcpy(dm.p, mu, sizeof(double) * nu_, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(dy.p, y, nu_, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(db.p, b, sizeof(double) * (size_t)(C + 1) * u, hipMemcpyHostToDevice));
  GP_HIP(hipMemcpy(du.p, unif, sizeof(double) * (size_t)(max_rounds + 2) * u, hipMemcpyHostToDevice));
  EssFArgs a{};
  a.f = df.as<double>(); a.nu = dn.as<double>(); a.ld = n; a.mu = dm.as<double>(); a.ld_mu = n;
  a.y = dy.as<int8_t>(); a.ldy = n; a.bthr = db.as<double>(); a.unif = du.as<double>();
  a.Np = n; a.U_pad = u; a.C = C; a.mloc = u; a.ML = u; a.m = u; a.R = max_rounds;
  a.rounds = dr.as<int32_t>(); a.ll = dl.as<double>();
  GP_TRY(launch_ess_f(0, a));
  GP_HIP(hipMemcpy(f, df.p, sizeof(double) * nu_, hipMemcpyDeviceToHost));
  if (rounds) GP_HIP(hipMemcpy(rounds, dr.p, sizeof(int32_t) * u, hipMemcpyDeviceToHost));
  if (ll) GP_HIP(hipMemcpy(ll, dl.p, sizeof(double) * u, hipMemcpyDeviceToHost));
  GP_HIP(hipDeviceSynchronize());
  return GPIRT_OK;
}

int gpirt_fstar_predict(const double *x, const double *L, const double *f, int32_t n, int32_t u,
                        double ls, double sf, int32_t n_grid, double grid_lo, double grid_step,
                        double *fstar) {
  if (!x || !L || !f || !fstar || n < 1 || u < 1 || n_grid < 1) {
    set_error("gpirt_fstar_predict: bad argument");
    return GPIRT_ERR_BAD_ARG;
  }
  GP_TRY(have_device());
  const int np = (int)pad_up(n), up = (int)pad_up(u), gp = (int)pad_up(n_grid);
  DevBuf dS, dI, dx, dW, dF, dGt, dout, dX;
  GP_TRY(factor_from_host_L(L, n, np, dS, dI));
  GP_TRY(dX.alloc(sizeof(double) * (size_t)4 * SUP * gp));
  GP_TRY(dx.alloc(sizeof(double) * n));
  GP_TRY(dW.alloc(sizeof(double) * (size_t)np * gp));
  GP_TRY(dF.alloc(sizeof(double) * (size_t)np * up));
  GP_TRY(dGt.alloc(sizeof(double) * (size_t)up * gp));
  GP_TRY(dout.alloc(sizeof(double) * (size_t)n_grid * u));
  GP_HIP(hipMemcpy(dx.p, x, sizeof(double) * n, hipMemcpyHostToDevice));
  GP_TRY(upload_padded(f, n, u, dF.as<double>(), np, 0));
  GP_TRY(launch_kstar_t(0, dx.as<double>(), 0, n, np, 1, ls, sf, n_grid, gp, grid_lo, grid_step,
                        dW.as<double>()));
  GP_TRY(launch_potrs(0, dS.as<double>(), dI.as<double>(), np, 1, dW.as<double>(), np, 0, gp,
                      dX.as<double>()));
  GemmArgs g{};
  g.A = dF.as<double>(); g.lda = np; g.B = dW.as<double>(); g.ldb = np; g.C = dGt.as<double>();
  g.ldc = up; g.M = up; g.N = gp; g.K = np; g.batch = 1; g.alpha = 1.0; g.beta = 0.0;
  GP_TRY(launch_gemm(0, true, false, g));
  GP_TRY(launch_export_fstar(0, dGt.as<double>(), up, n_grid, 1, u, up, dout.as<double>()));
  GP_HIP(hipMemcpy(fstar, dout.p, sizeof(double) * (size_t)n_grid * u, hipMemcpyDeviceToHost));
  GP_HIP(hipDeviceSynchronize());
  return GPIRT_OK;
}

}  // extern "C"

// ---- Cholesky without allocation ---------------------------------------------------------
// gpirt_potrf's device work needs three buffers beyond the caller's matrix: the matrix padded to
// tiles (pad block = identity), the inverted diagonal blocks the panel solves multiply by, and
// the status word.  Here they live in one file-scope device array, so that the entry point below
// never allocates.  In host code the array's name is only a handle for hipGetSymbolAddress: each
// GPU holds its own copy at its own address, looked up once per device and kept in g_scratch_ptr.
namespace {
constexpr int kNoAllocMaxN = GPIRT_POTRF_NOALLOC_MAX_N;                       // 1024
constexpr int64_t kScrS = (int64_t)kNoAllocMaxN * kNoAllocMaxN;               // padded matrix
constexpr int64_t kScrInv = ((kNoAllocMaxN / TILE + 3) / 4) * (int64_t)SUP * SUP;  // inverted blocks
constexpr int64_t kScrInfo = 32;                                              // status word (+ pad)
constexpr int64_t kScrRaw = POTRF_RAW_ELEMS;                                  // raw panels of the steps inside a super-panel
constexpr int64_t kScrElems = kScrS + kScrInv + kScrInfo + kScrRaw;
constexpr int kMaxDevices = 64;
}  // namespace
static __device__ __align__(256) double scratch[kScrElems];
static double *g_scratch_ptr[kMaxDevices];
static std::mutex g_scratch_mutex;

// the calling thread's current device's copy of `scratch`
static int scratch_for_current_device(double **out) {
  int dev = -1;
  GP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) { set_error("device index %d out of range", dev); return GPIRT_ERR_BAD_ARG; }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (!g_scratch_ptr[dev]) {
    double *p = nullptr;
    GP_HIP(hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch)));
    if (!p) { set_error("hipGetSymbolAddress(scratch) returned a null address"); return GPIRT_ERR_HIP; }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return GPIRT_OK;
}

// dst [np, np] <- src [n, n] (ld lds), zero pad rows / columns, identity on the pad block
__global__ void pad_in_kernel(const double *__restrict__ src, int64_t lds, int n, double *__restrict__ dst, int np) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= np) return;
  dst[i + (int64_t)np * j] = (i < n && j < n) ? src[i + lds * j] : (i == j ? 1.0 : 0.0);
}
// dst [n, n] (ld ldd) <- lower triangle of src [np, np], strict upper triangle zero
__global__ void tril_out_kernel(const double *__restrict__ src, int np, int n, double *__restrict__ dst, int64_t ldd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n) return;
  dst[i + ldd * j] = i >= j ? src[i + (int64_t)np * j] : 0.0;
}

extern "C" {

int gpirt_potrf_noalloc(const double *d_S, int64_t lds, int32_t n, double *d_L, int64_t ldl, void *stream) {
  if (!d_S || !d_L || n < 1 || lds < n || ldl < n) { set_error("gpirt_potrf_noalloc: bad argument"); return GPIRT_ERR_BAD_ARG; }
  if (n > kNoAllocMaxN) {
    set_error("gpirt_potrf_noalloc: n = %d exceeds the static scratch's limit of %d rows (use gpirt_potrf, "
              "which allocates)", n, kNoAllocMaxN);
    return GPIRT_ERR_BAD_ARG;
  }
  GP_TRY(have_device());
  double *base = nullptr;
  GP_TRY(scratch_for_current_device(&base));
  hipStream_t st = (hipStream_t)stream;
  const int np = (int)pad_up(n);
  double *S = base, *inv = base + kScrS;
  int *info = reinterpret_cast<int *>(base + kScrS + kScrInv);
  double *raw = base + kScrS + kScrInv + kScrInfo;      // written before it is read in every call
  // the array keeps the last call's contents: the inverted blocks' upper halves are never
  // written and must read zero, the status word must start clear, S is overwritten whole
  GP_HIP(hipMemsetAsync(inv, 0, sizeof(double) * (size_t)inv_group_elems(np), st));
  GP_HIP(hipMemsetAsync(info, 0, sizeof(int), st));
  hipLaunchKernelGGL(pad_in_kernel, dim3((np + 255) / 256, np), dim3(256), 0, st, d_S, lds, n, S, np);
  GP_HIP(hipGetLastError());
  GP_TRY(launch_potrf(st, S, np, 1, inv, info, raw, nullptr, POTRF_DEFER_INVERSES | POTRF_SUBSTITUTION));
  hipLaunchKernelGGL(tril_out_kernel, dim3((n + 255) / 256, n), dim3(256), 0, st, S, np, n, d_L, ldl);
  GP_HIP(hipGetLastError());
  int h_info = 0;
  GP_HIP(hipMemcpyAsync(&h_info, info, sizeof(int), hipMemcpyDeviceToHost, st));
  GP_HIP(hipStreamSynchronize(st));
  if (h_info) { set_error("gpirt_potrf_noalloc: pivot %d is not positive", h_info); return GPIRT_ERR_NOT_PD; }
  return GPIRT_OK;
}

static int rng_entry(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit, int64_t count,
                     int normal, double *out) {
  if (!out || count < 0) { set_error("gpirt_rng: bad argument"); return GPIRT_ERR_BAD_ARG; }
  GP_TRY(have_device());
  DevBuf d;
  GP_TRY(d.alloc(sizeof(double) * (size_t)count));
  GP_TRY(launch_rng_fill(0, seed, tag, sweep, unit, count, normal, d.as<double>()));
  GP_HIP(hipMemcpy(out, d.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost));
  return GPIRT_OK;
}

int gpirt_rng_normal(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit, int64_t count,
                     double *out) {
  return rng_entry(seed, tag, sweep, unit, count, 1, out);
}

int gpirt_rng_uniform(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit, int64_t count,
                      double *out) {
  return rng_entry(seed, tag, sweep, unit, count, 0, out);
}

}  // extern "C"
