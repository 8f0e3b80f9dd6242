/*
 * gpirt_oracle.c -- CPU restatement of the GPIRT Gibbs sweep.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker the HIP path is compared against and the CPU baseline the
 * benchmark times.  Nothing in the product path may call it.
 *
 * PARITY UNPINNED: the reference's production sampler is an external Rcpp package that is
 * not in the reference tree and the tree holds no expected-output vectors.  What this file
 * follows is the in-tree MATLAB prototype of the same sampler and the R call sites:
 *   sweep structure           main.m:46-87
 *   elliptical slice update   utils/elliptical_slice.m:47-102
 *   one item update           utils/sample_fs.m:1-6
 *   log-likelihood            utils/log_like_fn.m:3-4, link utils/sigmoid_fn.m:4 (probit),
 *                             ordinal form gpirt-sythetic.R:159-167, generate_data.R:73-80
 *   f* on the grid            main.m:55-60 (gp posterior mean), kernel utils/model.m:4-6,
 *                             getprob_gpirt.R:41-51
 *   respondent step           utils/sample_score.m:4-19 (grid inverse-CDF variant); the
 *                             slice variant looks f* up at the nearest grid point as
 *                             main.m:69 does
 *   state snap                main.m:69-73
 *   kept draws                main.m:41-43,90-92; gpirt-sythetic.R:111,123,153-154,193
 * Its deterministic pieces are pinned in tests/ against scipy (cholesky, ndtr,
 * solve_triangular) and the Philox4x32-10 known-answer vectors.
 *
 * Build: gcc -O3 -march=native -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include "../include/gpirt.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_API __attribute__((visibility("default")))

/* stream tags of the shared counter-based generator */
enum {
  TAG_F_Z = 1, TAG_F_U = 2, TAG_P_Z = 3, TAG_P_U = 4, TAG_T_Z = 5, TAG_T_U = 6, TAG_INIT = 7
};

/* ------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al., SC'11).  counter = (index, unit, sweep, tag), key = seed.
 * ---------------------------------------------------------------------------------- */
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
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

static inline double u53(uint32_t hi, uint32_t lo) {
  /* 53 random bits, centred so that 0 and 1 are never returned */
  return ((double)(hi >> 5) * 67108864.0 + (double)(lo >> 6) + 0.5) * (1.0 / 9007199254740992.0);
}

/* two uniforms of stream (seed,tag,sweep,unit) at pair index q */
static inline void rng_uniform_pair(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                                    uint32_t q, double *u0, double *u1) {
  uint32_t c[4] = {q, unit, sweep, tag};
  philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  *u0 = u53(c[0], c[1]);
  *u1 = u53(c[2], c[3]);
}

static inline double rng_uniform_at(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                                    uint32_t d) {
  double a, b;
  rng_uniform_pair(seed, tag, sweep, unit, d >> 1, &a, &b);
  return (d & 1) ? b : a;
}

/* Box-Muller on one pair */
static inline void rng_normal_pair(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                                   uint32_t q, double *z0, double *z1) {
  double a, b;
  rng_uniform_pair(seed, tag, sweep, unit, q, &a, &b);
  double r = sqrt(-2.0 * log(a));
  double t = 6.283185307179586476925286766559 * b;
  *z0 = r * cos(t);
  *z1 = r * sin(t);
}

static inline double rng_normal_at(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                                   uint32_t d) {
  double a, b;
  rng_normal_pair(seed, tag, sweep, unit, d >> 1, &a, &b);
  return (d & 1) ? b : a;
}

ORC_API int orc_rng_normal(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                           int64_t count, double *out) {
  for (int64_t d = 0; d < count; ++d) out[d] = rng_normal_at(seed, tag, sweep, unit, (uint32_t)d);
  return 0;
}

ORC_API int orc_rng_uniform(uint64_t seed, uint32_t tag, uint32_t sweep, uint32_t unit,
                            int64_t count, double *out) {
  for (int64_t d = 0; d < count; ++d) out[d] = rng_uniform_at(seed, tag, sweep, unit, (uint32_t)d);
  return 0;
}

/* raw Philox block, for the known-answer test */
ORC_API void orc_philox_raw(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4]) {
  uint32_t c[4] = {ctr[0], ctr[1], ctr[2], ctr[3]};
  philox4x32_10(c, key[0], key[1]);
  memcpy(out, c, sizeof c);
}

/* ------------------------------------------------------------------------------------
 * Probit link and the ordinal cell (sigmoid_fn.m:4; gpirt-sythetic.R:159-167)
 * ---------------------------------------------------------------------------------- */
#define INV_SQRT2 0.70710678118654752440084436210485
#define P_FLOOR 1e-300

/* log P(y | g) = log( Phi(b[y]-g) - Phi(b[y-1]-g) ), evaluated on the upper tails when
 * both arguments are positive so that it agrees with the naive difference wherever the
 * naive difference is finite, and stays finite where the naive one cancels to zero. */
static inline double ordinal_cell(int y, double g, const double *b) {
  double zlo = b[y - 1] - g, zhi = b[y] - g;
  double p;
  if (zlo > 0.0)
    p = 0.5 * erfc(zlo * INV_SQRT2) - 0.5 * erfc(zhi * INV_SQRT2);
  else
    p = 0.5 * erfc(-zhi * INV_SQRT2) - 0.5 * erfc(-zlo * INV_SQRT2);
  if (!(p > P_FLOOR)) p = P_FLOOR;
  return log(p);
}

ORC_API double orc_ordinal_cell(int y, double g, const double *b) { return ordinal_cell(y, g, b); }

ORC_API int orc_ordinal_loglik(const double *g, const int8_t *y, const double *b, int32_t n,
                               int32_t u, int32_t C, double *ll_unit, double *ll_row) {
  if (ll_row) memset(ll_row, 0, sizeof(double) * (size_t)n);
  for (int k = 0; k < u; ++k) {
    double bb[64];
    for (int c = 0; c <= C; ++c) bb[c] = b[c + (size_t)(C + 1) * k];
    bb[0] = -INFINITY; bb[C] = INFINITY;
    double s = 0.0;
    for (int i = 0; i < n; ++i) {
      int yy = y[i + (size_t)n * k];
      if (yy <= 0) continue;
      double v = ordinal_cell(yy, g[i + (size_t)n * k], bb);
      s += v;
      if (ll_row) ll_row[i] += v;
    }
    if (ll_unit) ll_unit[k] = s;
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * Dense pieces: kernel matrix, Cholesky, L*Z, solves
 * ---------------------------------------------------------------------------------- */
ORC_API int orc_k_build(const double *x, int32_t n, double ls, double sf, double jitter,
                        double *K) {
  double sf2 = sf * sf, inv = 1.0 / (ls * ls);
#pragma omp parallel for schedule(static)
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < n; ++i) {
      double d = x[i] - x[j];
      K[i + (size_t)n * j] = sf2 * exp(-0.5 * d * d * inv) + (i == j ? jitter : 0.0);
    }
  return 0;
}

/* in-place lower Cholesky, right-looking by columns; returns 0 or 1+index of the bad pivot */
static int potrf_lower_inplace(double *A, int n) {
  for (int j = 0; j < n; ++j) {
    double d = A[j + (size_t)n * j];
    if (!(d > 0.0)) return j + 1;
    d = sqrt(d);
    double *cj = A + (size_t)n * j;
    cj[j] = d;
    double inv = 1.0 / d;
    for (int i = j + 1; i < n; ++i) cj[i] *= inv;
#pragma omp parallel for schedule(dynamic, 8) if (n - j > 256)
    for (int k = j + 1; k < n; ++k) {
      double a = cj[k];
      double *ck = A + (size_t)n * k;
      for (int i = k; i < n; ++i) ck[i] -= a * cj[i];
    }
  }
  for (int j = 1; j < n; ++j)
    for (int i = 0; i < j; ++i) A[i + (size_t)n * j] = 0.0;
  return 0;
}

ORC_API int orc_potrf(const double *S, int32_t n, double *L) {
  if (L != S) memcpy(L, S, sizeof(double) * (size_t)n * n);
  return potrf_lower_inplace(L, n) ? GPIRT_ERR_NOT_PD : 0;
}

ORC_API int orc_trmm_lz(const double *L, const double *Z, int32_t n, int32_t k, double *N) {
#pragma omp parallel for schedule(static)
  for (int c = 0; c < k; ++c) {
    double *o = N + (size_t)n * c;
    const double *z = Z + (size_t)n * c;
    for (int i = 0; i < n; ++i) o[i] = 0.0;
    for (int j = 0; j < n; ++j) {
      double a = z[j];
      const double *lj = L + (size_t)n * j;
      for (int i = j; i < n; ++i) o[i] += lj[i] * a;
    }
  }
  return 0;
}

/* x <- S^-1 x for one column given the lower factor */
static void potrs_col(const double *L, int n, double *x) {
  for (int j = 0; j < n; ++j) { /* L w = x, column-oriented */
    const double *lj = L + (size_t)n * j;
    double w = x[j] / lj[j];
    x[j] = w;
    for (int i = j + 1; i < n; ++i) x[i] -= lj[i] * w;
  }
  for (int j = n - 1; j >= 0; --j) { /* L' a = w, dot-product form */
    const double *lj = L + (size_t)n * j;
    double s = x[j];
    for (int i = j + 1; i < n; ++i) s -= lj[i] * x[i];
    x[j] = s / lj[j];
  }
}

ORC_API int orc_potrs(const double *L, const double *B, int32_t n, int32_t k, double *X) {
  if (X != B) memcpy(X, B, sizeof(double) * (size_t)n * k);
#pragma omp parallel for schedule(static)
  for (int c = 0; c < k; ++c) potrs_col(L, n, X + (size_t)n * c);
  return 0;
}

ORC_API int orc_fstar_predict(const double *x, const double *L, const double *f, int32_t n,
                              int32_t u, double ls, double sf, int32_t G, double lo, double step,
                              double *fstar) {
  double *Ks = (double *)malloc(sizeof(double) * (size_t)G * n); /* [G,n] */
  double *al = (double *)malloc(sizeof(double) * (size_t)n * u);
  if (!Ks || !al) { free(Ks); free(al); return GPIRT_ERR_BAD_ARG; }
  double sf2 = sf * sf, inv = 1.0 / (ls * ls);
#pragma omp parallel for schedule(static)
  for (int r = 0; r < n; ++r)
    for (int g = 0; g < G; ++g) {
      double d = (lo + step * g) - x[r];
      Ks[g + (size_t)G * r] = sf2 * exp(-0.5 * d * d * inv);
    }
  orc_potrs(L, f, n, u, al);
#pragma omp parallel for schedule(static)
  for (int k = 0; k < u; ++k) {
    double *o = fstar + (size_t)G * k;
    const double *a = al + (size_t)n * k;
    for (int g = 0; g < G; ++g) o[g] = 0.0;
    for (int r = 0; r < n; ++r) {
      double ar = a[r];
      const double *kr = Ks + (size_t)G * r;
      for (int g = 0; g < G; ++g) o[g] += kr[g] * ar;
    }
  }
  free(Ks); free(al);
  return 0;
}

/* ------------------------------------------------------------------------------------
 * Elliptical slice update with injected randomness (utils/elliptical_slice.m:47-102)
 * ---------------------------------------------------------------------------------- */
static double ll_f_unit(const double *f, const double *mu, const int8_t *y, const double *b,
                        int n) {
  double s = 0.0;
  for (int i = 0; i < n; ++i)
    if (y[i] > 0) s += ordinal_cell(y[i], (mu ? mu[i] : 0.0) + f[i], b);
  return s;
}

#define TWO_PI 6.283185307179586476925286766559

ORC_API int orc_ess_f(double *f, const double *nu, const double *mu, const int8_t *y,
                      const double *b, const double *unif, int32_t n, int32_t u, int32_t C,
                      int32_t max_rounds, int32_t *rounds, double *ll) {
#pragma omp parallel for schedule(dynamic)
  for (int k = 0; k < u; ++k) {
    double bb[64];
    for (int c = 0; c <= C; ++c) bb[c] = b[c + (size_t)(C + 1) * k];
    bb[0] = -INFINITY; bb[C] = INFINITY;
    double *fk = f + (size_t)n * k;
    const double *nk = nu + (size_t)n * k;
    const double *mk = mu ? mu + (size_t)n * k : NULL;
    const int8_t *yk = y + (size_t)n * k;
    const double *uk = unif + (size_t)(max_rounds + 2) * k;
    double *prop = (double *)malloc(sizeof(double) * (size_t)n);
    double cur = ll_f_unit(fk, mk, yk, bb, n);
    double hh = log(uk[0]) + cur;                       /* :66 */
    double phi = uk[1] * TWO_PI;                        /* :72 */
    double pmin = phi - TWO_PI, pmax = phi;             /* :73-74 */
    int r = 0, ok = 0;
    double l = cur;
    while (r < max_rounds) {
      double c = cos(phi), s = sin(phi);
      for (int i = 0; i < n; ++i) prop[i] = fk[i] * c + nk[i] * s; /* :85 */
      l = ll_f_unit(prop, mk, yk, bb, n);
      ++r;
      if (l > hh) { ok = 1; break; }                    /* :87 */
      if (phi > 0.0) pmax = phi;                        /* :92-96 */
      else if (phi < 0.0) pmin = phi;
      else break;
      phi = uk[1 + r] * (pmax - pmin) + pmin;           /* :100 */
    }
    if (ok) { memcpy(fk, prop, sizeof(double) * (size_t)n); if (ll) ll[k] = l; }
    else if (ll) ll[k] = cur;
    if (rounds) rounds[k] = ok ? r : -r;
    free(prop);
  }
  return 0;
}

/* ------------------------------------------------------------------------------------
 * The sampler
 * ---------------------------------------------------------------------------------- */
typedef struct orc_sampler {
  gpirt_config cfg;
  int n, m, H, C, P, Np, mloc, U, G, j0;
  int fixed_thr;
  uint64_t seed;
  uint32_t sweep;
  int8_t *yu;      /* [Np,U] */
  double *theta;   /* [n,H]  */
  double *f;       /* [Np,U] */
  double *ebeta;   /* [3,U] whitened (beta-mean)/sd */
  double *delta;   /* [C-1,U] */
  double *beta;    /* [3,U]  */
  double *bthr;    /* [C+1,U]*/
  double *bpm, *bps; /* [3,mloc] */
  double *tpm, *tps; /* [2,n] */
  double *L;       /* [Np,Np,P] */
  double *fstar;   /* [G,U] */
  double *Lt;      /* [H,H] chol of temporal kernel */
  double jitter_now;
  int64_t r_f, r_p, r_t, capped;
  int jitter_retries;
  double last_ll;
  double phase_s[7]; /* factor, nu, ess_f, param, fstar, theta, snap */
} orc_sampler;

static void params_from_white(const orc_sampler *s, int u, const double *e, const double *dl,
                              double *beta, double *b) {
  int jl = u % s->mloc;
  for (int k = 0; k < 3; ++k) beta[k] = s->bpm[k + 3 * jl] + s->bps[k + 3 * jl] * e[k];
  int C = s->C;
  b[0] = -INFINITY; b[C] = INFINITY;
  if (C >= 2) {
    b[1] = dl[0];
    for (int c = 2; c < C; ++c) b[c] = b[c - 1] + exp(dl[c - 1]);
  }
}

static inline int grid_index(const orc_sampler *s, double x) {
  double t = (x - s->cfg.grid_lo) / s->cfg.grid_step;
  double r = (s->cfg.flags & GPIRT_INDEX_TRUNCATE) ? floor(t) : rint(t);   /* gpirt-sythetic.R:190 : main.m:69 */
  if (!(r > 0.0)) return 0;
  if (r > (double)(s->G - 1)) return s->G - 1;
  return (int)r;
}

static double temporal_k(const gpirt_config *c, double d) {
  if (c->theta_os == 0.0) return (c->theta_ls < 0.0) ? 1.0 : (d == 0.0 ? 1.0 : 0.0);
  double os2 = c->theta_os * c->theta_os;
  if (c->theta_ls <= 0.0) return os2;
  double a = fabs(d) / c->theta_ls;
  if (c->kernel == GPIRT_KERNEL_MATERN52) {
    double q = 2.2360679774997896964 * a;
    return os2 * (1.0 + q + q * q / 3.0) * exp(-q);
  }
  return os2 * exp(-0.5 * a * a);
}

ORC_API void orc_sampler_destroy(orc_sampler *s) {
  if (!s) return;
  free(s->yu); free(s->theta); free(s->f); free(s->ebeta); free(s->delta); free(s->beta);
  free(s->bthr); free(s->bpm); free(s->bps); free(s->tpm); free(s->tps); free(s->L);
  free(s->fstar); free(s->Lt); free(s);
}

ORC_API int orc_sampler_create(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
                               const double *bpm, const double *bps, const double *tpm,
                               const double *tps, const double *thresholds, orc_sampler **out) {
  if (!cfg || !y || !out || cfg->n < 1 || cfg->m < 1 || cfg->H < 1 || cfg->C < 2 || cfg->C > 60)
    return GPIRT_ERR_BAD_ARG;
  orc_sampler *s = (orc_sampler *)calloc(1, sizeof *s);
  s->cfg = *cfg;
  s->n = cfg->n; s->m = cfg->m; s->H = cfg->H; s->C = cfg->C; s->G = cfg->n_grid;
  s->P = cfg->constant_irf ? 1 : cfg->H;
  s->Np = cfg->constant_irf ? cfg->n * cfg->H : cfg->n;
  s->j0 = cfg->item_begin;
  int j1 = cfg->item_end > cfg->item_begin ? cfg->item_end : cfg->m;
  if (cfg->item_end <= cfg->item_begin) s->j0 = 0;
  s->mloc = j1 - s->j0;
  s->U = s->mloc * s->P;
  s->seed = cfg->seed + 0x9E3779B97F4A7C15ull * (uint64_t)cfg->chain;
  s->jitter_now = cfg->jitter;
  if (s->cfg.max_ess_rounds <= 0) s->cfg.max_ess_rounds = 200;
  if (cfg->theta_mode == GPIRT_THETA_GRID && cfg->H != 1) { free(s); return GPIRT_ERR_BAD_ARG; }
  int n = s->n, m = s->m, H = s->H, C = s->C, Np = s->Np, U = s->U, mloc = s->mloc, P = s->P;
  s->yu = (int8_t *)malloc((size_t)Np * U);
  s->theta = (double *)malloc(sizeof(double) * (size_t)n * H);
  s->f = (double *)calloc((size_t)Np * U, sizeof(double));
  s->ebeta = (double *)calloc((size_t)3 * U, sizeof(double));
  s->delta = (double *)calloc((size_t)(C - 1) * U, sizeof(double));
  s->beta = (double *)calloc((size_t)3 * U, sizeof(double));
  s->bthr = (double *)calloc((size_t)(C + 1) * U, sizeof(double));
  s->bpm = (double *)calloc((size_t)3 * mloc, sizeof(double));
  s->bps = (double *)calloc((size_t)3 * mloc, sizeof(double));
  s->tpm = (double *)calloc((size_t)2 * n, sizeof(double));
  s->tps = (double *)calloc((size_t)2 * n, sizeof(double));
  s->L = (double *)malloc(sizeof(double) * (size_t)Np * Np * P);
  s->fstar = (double *)calloc((size_t)s->G * U, sizeof(double));
  s->Lt = (double *)calloc((size_t)H * H, sizeof(double));
  for (int jl = 0; jl < mloc; ++jl)
    for (int k = 0; k < 3; ++k) {
      s->bpm[k + 3 * jl] = bpm ? bpm[k + 3 * (s->j0 + jl)] : 0.0;
      s->bps[k + 3 * jl] = bps ? bps[k + 3 * (s->j0 + jl)] : 1.0;
    }
  if (tpm) memcpy(s->tpm, tpm, sizeof(double) * 2 * (size_t)n);
  if (tps) memcpy(s->tps, tps, sizeof(double) * 2 * (size_t)n);
  /* responses, unit-major */
  for (int h = 0; h < H; ++h)
    for (int jl = 0; jl < mloc; ++jl)
      for (int i = 0; i < n; ++i) {
        int8_t v = y[i + (size_t)n * ((s->j0 + jl) + (size_t)m * h)];
        if (v < 0 || v > C) v = 0;
        size_t u = cfg->constant_irf ? (size_t)jl : (size_t)jl + (size_t)mloc * h;
        size_t r = cfg->constant_irf ? (size_t)i + (size_t)n * h : (size_t)i;
        s->yu[r + (size_t)Np * u] = v;
      }
  /* theta */
  for (int h = 0; h < H; ++h)
    for (int i = 0; i < n; ++i)
      s->theta[i + (size_t)n * h] =
          theta_init ? theta_init[i + (size_t)n * h]
                     : rng_normal_at(s->seed, TAG_INIT, 0, (uint32_t)i, 0);
  /* thresholds */
  s->fixed_thr = thresholds != NULL;
  for (int u = 0; u < U; ++u) {
    double *dl = s->delta + (size_t)(C - 1) * u;
    double b[64];
    for (int c = 1; c < C; ++c) {
      if (thresholds) b[c] = thresholds[c];
      else b[c] = (C == 2) ? 0.0 : -1.5 + 3.0 * (double)(c - 1) / (double)(C - 2);
    }
    dl[0] = b[1];
    for (int c = 2; c < C; ++c) dl[c - 1] = log(b[c] - b[c - 1]);
    params_from_white(s, u, s->ebeta + 3 * (size_t)u, dl, s->beta + 3 * (size_t)u,
                      s->bthr + (size_t)(C + 1) * u);
  }
  /* temporal prior factor: kernel over t = 1..H plus the generator's 1e-6 jitter
   * (generate_data.R:54-56) */
  for (int a = 0; a < H; ++a)
    for (int b2 = 0; b2 < H; ++b2)
      s->Lt[a + (size_t)H * b2] = temporal_k(cfg, (double)(a - b2)) + (a == b2 ? 1e-6 : 0.0);
  if (potrf_lower_inplace(s->Lt, H)) { orc_sampler_destroy(s); return GPIRT_ERR_NOT_PD; }
  *out = s;
  return 0;
}

/* ---- sweep phases ---- */
static int phase_factor(orc_sampler *s) {
  int Np = s->Np;
  for (;;) {
    int bad = 0;
    for (int p = 0; p < s->P && !bad; ++p) {
      double *Lp = s->L + (size_t)Np * Np * p;
      orc_k_build(s->theta + (size_t)Np * p, Np, s->cfg.irf_ls, s->cfg.irf_sf, s->jitter_now, Lp);
      if (potrf_lower_inplace(Lp, Np)) bad = 1;
    }
    if (!bad) return 0;
    if (s->jitter_now >= 1e-2) return GPIRT_ERR_NOT_PD;
    s->jitter_now *= 10.0;
    s->jitter_retries++;
  }
}

static inline uint32_t global_unit(const orc_sampler *s, int u) {
  int jl = u % s->mloc, p = u / s->mloc;
  return (uint32_t)((s->j0 + jl) + s->m * p);
}

static void unit_mu(const orc_sampler *s, int u, const double *beta, double *mu) {
  const double *x = s->theta + (size_t)s->Np * (u / s->mloc);
  for (int r = 0; r < s->Np; ++r) mu[r] = beta[0] + beta[1] * x[r] + beta[2] * x[r] * x[r];
}

static double now_s(void) {
#ifdef _OPENMP
  return omp_get_wtime();
#else
  return (double)clock() / CLOCKS_PER_SEC;
#endif
}

static void phase_f(orc_sampler *s) {
  int Np = s->Np, U = s->U, C = s->C, R = s->cfg.max_ess_rounds;
  int64_t rounds_tot = 0, capped = 0;
  double t0 = now_s();
  double *NU = (double *)malloc(sizeof(double) * (size_t)Np * U);
#pragma omp parallel for schedule(dynamic)
  for (int u = 0; u < U; ++u) {   /* nu_u = L z_u  (sample_fs.m:2) */
    uint32_t ug = global_unit(s, u);
    double *z = (double *)malloc(sizeof(double) * (size_t)Np);
    for (int r = 0; r < Np; ++r) z[r] = rng_normal_at(s->seed, TAG_F_Z, s->sweep, ug, (uint32_t)r);
    orc_trmm_lz(s->L + (size_t)Np * Np * (u / s->mloc), z, Np, 1, NU + (size_t)Np * u);
    free(z);
  }
  double t1 = now_s();
  s->phase_s[1] += t1 - t0;
#pragma omp parallel for schedule(dynamic) reduction(+ : rounds_tot, capped)
  for (int u = 0; u < U; ++u) {
    uint32_t ug = global_unit(s, u);
    double *mu = (double *)malloc(sizeof(double) * (size_t)Np);
    double *nu = NU + (size_t)Np * u;
    double *un = (double *)malloc(sizeof(double) * (size_t)(R + 2));
    unit_mu(s, u, s->beta + 3 * (size_t)u, mu);
    for (int d = 0; d < R + 2; ++d) un[d] = rng_uniform_at(s->seed, TAG_F_U, s->sweep, ug, (uint32_t)d);
    int32_t rr;
    orc_ess_f(s->f + (size_t)Np * u, nu, mu, s->yu + (size_t)Np * u,
              s->bthr + (size_t)(C + 1) * u, un, Np, 1, C, R, &rr, NULL);
    rounds_tot += 1 + (rr < 0 ? -rr : rr);
    if (rr < 0) capped++;
    free(mu); free(un);
  }
  free(NU);
  s->phase_s[2] += now_s() - t1;
  s->r_f += rounds_tot; s->capped += capped;
}

/* joint slice update of (beta, thresholds) of every unit: the vector is the whitened
 * coefficients e[3] and the unconstrained thresholds delta[C-1], prior N(0,I); a fixed
 * coordinate has zero prior variance, i.e. its ellipse component is 0. */
static double ll_param_unit(const orc_sampler *s, int u, const double *e, const double *dl) {
  double beta[3], b[64];
  params_from_white(s, u, e, dl, beta, b);
  const double *x = s->theta + (size_t)s->Np * (u / s->mloc);
  const double *f = s->f + (size_t)s->Np * u;
  const int8_t *y = s->yu + (size_t)s->Np * u;
  double acc = 0.0;
  for (int r = 0; r < s->Np; ++r)
    if (y[r] > 0)
      acc += ordinal_cell(y[r], beta[0] + beta[1] * x[r] + beta[2] * x[r] * x[r] + f[r], b);
  return acc;
}

static void phase_param(orc_sampler *s) {
  int U = s->U, C = s->C, R = s->cfg.max_ess_rounds, D = 3 + C - 1;
  int64_t rounds_tot = 0, capped = 0;
#pragma omp parallel for schedule(dynamic) reduction(+ : rounds_tot, capped)
  for (int u = 0; u < U; ++u) {
    uint32_t ug = global_unit(s, u);
    int jl = u % s->mloc;
    double v[64], nv[64], pv[64];
    double *e = s->ebeta + 3 * (size_t)u, *dl = s->delta + (size_t)(C - 1) * u;
    int anyfree = 0;
    for (int k = 0; k < D; ++k) {
      int is_free = k < 3 ? (s->bps[k + 3 * jl] > 0.0) : !s->fixed_thr;
      v[k] = k < 3 ? e[k] : dl[k - 3];
      nv[k] = is_free ? rng_normal_at(s->seed, TAG_P_Z, s->sweep, ug, (uint32_t)k) : 0.0;
      if (!is_free && k < 3) v[k] = 0.0;
      anyfree |= is_free;
    }
    if (!anyfree) continue;
    /* fixed thresholds keep their value: v stays, nv = 0, and cos(phi) must not scale
       them, so they are excluded from the rotation below */
    double cur = ll_param_unit(s, u, v, v + 3);
    double hh = log(rng_uniform_at(s->seed, TAG_P_U, s->sweep, ug, 0)) + cur;
    double phi = rng_uniform_at(s->seed, TAG_P_U, s->sweep, ug, 1) * TWO_PI;
    double pmin = phi - TWO_PI, pmax = phi;
    int r = 0, ok = 0;
    while (r < R) {
      double c = cos(phi), sn = sin(phi);
      for (int k = 0; k < D; ++k) {
        int rot = k < 3 ? 1 : !s->fixed_thr;
        pv[k] = rot ? v[k] * c + nv[k] * sn : v[k];
      }
      double l = ll_param_unit(s, u, pv, pv + 3);
      ++r;
      if (l > hh) { ok = 1; break; }
      if (phi > 0.0) pmax = phi; else if (phi < 0.0) pmin = phi; else break;
      phi = rng_uniform_at(s->seed, TAG_P_U, s->sweep, ug, (uint32_t)(1 + r)) * (pmax - pmin) + pmin;
    }
    if (ok) {
      for (int k = 0; k < 3; ++k) e[k] = pv[k];
      for (int k = 3; k < D; ++k) dl[k - 3] = pv[k];
      params_from_white(s, u, e, dl, s->beta + 3 * (size_t)u, s->bthr + (size_t)(C + 1) * u);
    } else capped++;
    rounds_tot += 1 + r;
  }
  s->r_p += rounds_tot; s->capped += capped;
}

static void phase_fstar(orc_sampler *s) {
  int Np = s->Np, mloc = s->mloc;
  for (int p = 0; p < s->P; ++p)
    orc_fstar_predict(s->theta + (size_t)Np * p, s->L + (size_t)Np * Np * p,
                      s->f + (size_t)Np * mloc * p, Np, mloc, s->cfg.irf_ls, s->cfg.irf_sf, s->G,
                      s->cfg.grid_lo, s->cfg.grid_step, s->fstar + (size_t)s->G * mloc * p);
}

/* local-items log-likelihood of respondent i at trajectory th[H], by grid lookup */
static double ll_resp(const orc_sampler *s, int i, const double *th) {
  double acc = 0.0;
  int n = s->n, mloc = s->mloc, C = s->C, G = s->G, Np = s->Np;
  for (int h = 0; h < s->H; ++h) {
    int g = grid_index(s, th[h]);
    double xg = s->cfg.grid_lo + s->cfg.grid_step * g;
    int p = s->cfg.constant_irf ? 0 : h;
    size_t r = s->cfg.constant_irf ? (size_t)i + (size_t)n * h : (size_t)i;
    for (int jl = 0; jl < mloc; ++jl) {
      size_t u = (size_t)jl + (size_t)mloc * p;
      int y = s->yu[r + (size_t)Np * u];
      if (y <= 0) continue;
      const double *be = s->beta + 3 * u;
      double gg = s->fstar[g + (size_t)G * u] + be[0] + be[1] * xg + be[2] * xg * xg;
      acc += ordinal_cell(y, gg, s->bthr + (size_t)(C + 1) * u);
    }
  }
  return acc;
}

static int allreduce(orc_sampler *s, double *buf, int64_t count) {
  if (s->cfg.world > 1) {
    if (!s->cfg.allreduce) return GPIRT_ERR_COMM;
    if (s->cfg.allreduce(buf, count, s->cfg.allreduce_user)) return GPIRT_ERR_COMM;
  }
  return 0;
}

/* respondent step, slice variant: lock-step rounds over all respondents; one n-vector
 * sum-all-reduce per round when items are sharded */
static int phase_theta_ess(orc_sampler *s) {
  int n = s->n, H = s->H, R = s->cfg.max_ess_rounds;
  double *x = (double *)malloc(sizeof(double) * (size_t)n * H * 3); /* centred state, nu, prop */
  double *nu = x + (size_t)n * H, *pr = nu + (size_t)n * H;
  double *hh = (double *)malloc(sizeof(double) * (size_t)n * 5);
  double *phi = hh + n, *pmin = phi + n, *pmax = pmin + n, *buf = pmax + n;
  char *done = (char *)calloc((size_t)n, 1);
  int rc = 0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    double z[130];
    for (int k = 0; k < H + 2; ++k) z[k] = rng_normal_at(s->seed, TAG_T_Z, s->sweep, (uint32_t)i, (uint32_t)k);
    for (int h = 0; h < H; ++h) {
      double t = (double)(h + 1);
      double acc = s->tps[0 + 2 * i] * z[H] + s->tps[1 + 2 * i] * t * z[H + 1];
      for (int k = 0; k <= h; ++k) acc += s->Lt[h + (size_t)H * k] * z[k];
      nu[i + (size_t)n * h] = acc;
      x[i + (size_t)n * h] = s->theta[i + (size_t)n * h] - (s->tpm[0 + 2 * i] + s->tpm[1 + 2 * i] * t);
    }
    double th[128];
    for (int h = 0; h < H; ++h) th[h] = s->theta[i + (size_t)n * h];
    buf[i] = ll_resp(s, i, th);
  }
  if ((rc = allreduce(s, buf, n))) goto out;
  s->r_t += 1;
  for (int i = 0; i < n; ++i) {
    hh[i] = log(rng_uniform_at(s->seed, TAG_T_U, s->sweep, (uint32_t)i, 0)) + buf[i];
    phi[i] = rng_uniform_at(s->seed, TAG_T_U, s->sweep, (uint32_t)i, 1) * TWO_PI;
    pmin[i] = phi[i] - TWO_PI; pmax[i] = phi[i];
  }
  for (int r = 1; r <= R; ++r) {
    int remaining = 0;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
      buf[i] = 0.0;
      if (done[i]) continue;
      double c = cos(phi[i]), sn = sin(phi[i]), th[128];
      for (int h = 0; h < H; ++h) {
        double t = (double)(h + 1);
        double v = x[i + (size_t)n * h] * c + nu[i + (size_t)n * h] * sn;
        pr[i + (size_t)n * h] = v;
        th[h] = v + (s->tpm[0 + 2 * i] + s->tpm[1 + 2 * i] * t);
      }
      buf[i] = ll_resp(s, i, th);
    }
    if ((rc = allreduce(s, buf, n))) goto out;
    s->r_t += 1;
    for (int i = 0; i < n; ++i) {
      if (done[i]) continue;
      if (buf[i] > hh[i]) {
        done[i] = 1;
        for (int h = 0; h < H; ++h)
          s->theta[i + (size_t)n * h] =
              pr[i + (size_t)n * h] + (s->tpm[0 + 2 * i] + s->tpm[1 + 2 * i] * (double)(h + 1));
        continue;
      }
      if (phi[i] > 0.0) pmax[i] = phi[i];
      else if (phi[i] < 0.0) pmin[i] = phi[i];
      else { done[i] = 1; s->capped++; continue; }
      phi[i] = rng_uniform_at(s->seed, TAG_T_U, s->sweep, (uint32_t)i, (uint32_t)(1 + r)) *
                   (pmax[i] - pmin[i]) + pmin[i];
      remaining++;
    }
    if (!remaining) break;
    if (r == R) s->capped += remaining;
  }
out:
  free(x); free(hh); free(done);
  return rc;
}

/* respondent step, grid inverse-CDF variant (utils/sample_score.m:4-19), H = 1.  The
 * prototype exponentiates unshifted log-posteriors (:10); the maximum is subtracted here
 * first, which leaves the normalised cdf of :11 unchanged. */
static int phase_theta_grid(orc_sampler *s) {
  int n = s->n, G = s->G, mloc = s->mloc, C = s->C, Np = s->Np;
  double *lp = (double *)calloc((size_t)n * G, sizeof(double)); /* [G,n] */
  int rc = 0;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i)
    for (int g = 0; g < G; ++g) {
      double xg = s->cfg.grid_lo + s->cfg.grid_step * g, acc = 0.0;
      for (int jl = 0; jl < mloc; ++jl) {
        int y = s->yu[i + (size_t)Np * jl];
        if (y <= 0) continue;
        const double *be = s->beta + 3 * (size_t)jl;
        acc += ordinal_cell(y, s->fstar[g + (size_t)G * jl] + be[0] + be[1] * xg + be[2] * xg * xg,
                            s->bthr + (size_t)(C + 1) * jl);
      }
      lp[g + (size_t)G * i] = acc;
    }
  if ((rc = allreduce(s, lp, (int64_t)n * G))) { free(lp); return rc; }
  s->r_t += 1;
  double os2 = s->cfg.theta_os > 0 ? s->cfg.theta_os * s->cfg.theta_os : 1.0;
  for (int i = 0; i < n; ++i) {
    double *l = lp + (size_t)G * i, mx = -INFINITY;
    double pv = os2 + s->tps[2 * i] * s->tps[2 * i] + s->tps[1 + 2 * i] * s->tps[1 + 2 * i];
    double pm = s->tpm[2 * i] + s->tpm[1 + 2 * i];
    for (int g = 0; g < G; ++g) {
      double xg = s->cfg.grid_lo + s->cfg.grid_step * g - pm;
      l[g] += -0.5 * xg * xg / pv;                     /* sample_score.m:7 up to a constant */
      if (l[g] > mx) mx = l[g];
    }
    double c = 0.0;
    for (int g = 0; g < G; ++g) { c += exp(l[g] - mx); l[g] = c; }   /* :10 */
    double c0 = l[0], span = l[G - 1] - l[0];
    double r = rng_uniform_at(s->seed, TAG_T_U, s->sweep, (uint32_t)i, 0);
    int pick = G - 1;                                  /* :12 */
    for (int g = 0; g < G; ++g)
      if ((l[g] - c0) / span > r) { pick = g; break; } /* :11,15-16 */
    s->theta[i] = s->cfg.grid_lo + s->cfg.grid_step * pick;
  }
  free(lp);
  return 0;
}

static void phase_snap(orc_sampler *s) {
  int Np = s->Np, G = s->G;
  for (int u = 0; u < s->U; ++u) {
    const double *x = s->theta + (size_t)Np * (u / s->mloc);
    for (int r = 0; r < Np; ++r)
      s->f[r + (size_t)Np * u] = s->fstar[grid_index(s, x[r]) + (size_t)G * u]; /* main.m:69-71 */
  }
}

static double total_loglik(const orc_sampler *s) {
  double acc = 0.0;
  for (int u = 0; u < s->U; ++u)
    acc += ll_param_unit(s, u, s->ebeta + 3 * (size_t)u, s->delta + (size_t)(s->C - 1) * u);
  return acc;
}

ORC_API int orc_sampler_sweep(orc_sampler *s, int32_t sweeps) {
  for (int it = 0; it < sweeps; ++it) {
    int rc;
    double t = now_s(), t2;
    if ((rc = phase_factor(s))) return rc;     /* main.m:67 (+ chol of sample_fs.m:2) */
    t2 = now_s(); s->phase_s[0] += t2 - t; t = t2;
    phase_f(s);                                 /* main.m:50-52 */
    t = now_s();
    phase_param(s);
    t2 = now_s(); s->phase_s[3] += t2 - t; t = t2;
    phase_fstar(s);                             /* main.m:55-60 */
    t2 = now_s(); s->phase_s[4] += t2 - t; t = t2;
    rc = s->cfg.theta_mode == GPIRT_THETA_GRID ? phase_theta_grid(s) : phase_theta_ess(s); /* :63 */
    if (rc) return rc;
    t2 = now_s(); s->phase_s[5] += t2 - t; t = t2;
    phase_snap(s);                              /* main.m:69-73 */
    s->phase_s[6] += now_s() - t;
    s->sweep++;
  }
  return 0;
}

ORC_API int orc_sampler_get_state(orc_sampler *s, double *theta, double *f, double *fstar,
                                  double *beta, double *threshold) {
  int n = s->n, H = s->H, mloc = s->mloc, C = s->C, G = s->G, P = s->P;
  if (theta) memcpy(theta, s->theta, sizeof(double) * (size_t)n * H);
  if (f)
    for (int h = 0; h < H; ++h)
      for (int jl = 0; jl < mloc; ++jl)
        for (int i = 0; i < n; ++i) {
          size_t u = s->cfg.constant_irf ? (size_t)jl : (size_t)jl + (size_t)mloc * h;
          size_t r = s->cfg.constant_irf ? (size_t)i + (size_t)n * h : (size_t)i;
          f[i + (size_t)n * (jl + (size_t)mloc * h)] = s->f[r + (size_t)s->Np * u];
        }
  if (fstar) memcpy(fstar, s->fstar, sizeof(double) * (size_t)G * s->U);
  if (beta) memcpy(beta, s->beta, sizeof(double) * 3 * (size_t)s->U);
  if (threshold) /* [mloc, C+1, P] */
    for (int p = 0; p < P; ++p)
      for (int c = 0; c <= C; ++c)
        for (int jl = 0; jl < mloc; ++jl)
          threshold[jl + (size_t)mloc * (c + (size_t)(C + 1) * p)] =
              s->bthr[c + (size_t)(C + 1) * (jl + (size_t)mloc * p)];
  return 0;
}

ORC_API int orc_sampler_diagnostics(orc_sampler *s, gpirt_outputs *o) {
  o->ess_f_rounds = s->r_f; o->ess_param_rounds = s->r_p; o->ess_theta_rounds = s->r_t;
  o->ess_capped = s->capped; o->jitter_retries = s->jitter_retries;
  return 0;
}

ORC_API double orc_sampler_loglik(orc_sampler *s) { return total_loglik(s); }

ORC_API void orc_sampler_phase_seconds(orc_sampler *s, double *out) {
  memcpy(out, s->phase_s, sizeof s->phase_s);
}

ORC_API int orc_run(const gpirt_config *cfg, const int8_t *y, const double *theta_init,
                    const double *bpm, const double *bps, const double *tpm, const double *tps,
                    const double *thresholds, gpirt_outputs *out) {
  orc_sampler *s = NULL;
  int rc = orc_sampler_create(cfg, y, theta_init, bpm, bps, tpm, tps, thresholds, &s);
  if (rc) return rc;
  int total = cfg->burn_iterations + cfg->sample_iterations;
  int thin = cfg->thin > 0 ? cfg->thin : 1;
  int K = cfg->sample_iterations / thin, k = 0;
  size_t nH = (size_t)s->n * s->H, sf = (size_t)s->n * s->mloc * s->H,
         sfs = (size_t)s->G * s->U, sb = 3 * (size_t)s->U, st = (size_t)(s->C + 1) * s->U;
  for (int it = 0; it < total; ++it) {
    if ((rc = orc_sampler_sweep(s, 1))) break;
    if (out && out->loglik) out->loglik[it] = total_loglik(s);
    int post = it - cfg->burn_iterations;
    if (out && post >= 0 && (post + 1) % thin == 0 && k < K) {
      orc_sampler_get_state(s, out->theta ? out->theta + nH * k : NULL,
                            (out->f && (cfg->flags & GPIRT_RETURN_F)) ? out->f + sf * k : NULL,
                            (out->fstar && (cfg->flags & GPIRT_RETURN_FSTAR)) ? out->fstar + sfs * k : NULL,
                            out->beta ? out->beta + sb * k : NULL,
                            out->threshold ? out->threshold + st * k : NULL);
      ++k;
    }
    if (cfg->progress && cfg->progress(it + 1, total, cfg->progress_user)) {
      rc = GPIRT_ERR_INTERRUPT; break;
    }
  }
  if (out) orc_sampler_diagnostics(s, out);
  orc_sampler_destroy(s);
  return rc;
}

ORC_API int orc_num_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
