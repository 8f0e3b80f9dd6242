/*
 * gpirt_shim.c -- `.Call` bridge from R to the C ABI of include/gpirt.h.
 *
 * Replaces the Rcpp entry of the external `gpirt` package behind gpirtMCMC()
 * (call sites: gpirt-sythetic.R:88-94, gpirt-senate.R:77-81, TAPS/gpirt_TAPS.R:30-36).
 * Build, on a machine with R:   R CMD SHLIB gpirt_shim.c -I../include -L../gpirt_amd -lgpirt_hip
 * NOT compiled in this repository's build: R is not available in the build image.
 *
 * All work happens on the R main thread.  Between sweeps the library calls the progress hook,
 * which asks R whether the user has interrupted -- inside R_ToplevelExec, so that R's long jump
 * stays within the hook -- and returns non-zero if so: gpirt_run then frees the resident sampler
 * and returns GPIRT_ERR_INTERRUPT, and only after that does this shim raise the R error.  No jump
 * ever crosses the library's C++ frames and nothing on the device is leaked by Ctrl-C.
 */
#include <R.h>
#include <Rinternals.h>
#include <R_ext/Utils.h>
#include <stdint.h>
#include <string.h>

#include "gpirt.h"

static void check_interrupt(void *unused) { (void)unused; R_CheckUserInterrupt(); }

static int progress_cb(int32_t sweep, int32_t total, void *user) {
  (void)sweep; (void)total; (void)user;
  /* R_ToplevelExec returns FALSE when the function left through a long jump (an interrupt) */
  return R_ToplevelExec(check_interrupt, NULL) == FALSE;
}

static const double *opt_real(SEXP x) { return Rf_isNull(x) ? NULL : REAL(x); }

/* .Call("gpirt_run_call", y_int, dims, iters, hyper, theta_init, bpm, bps, tpm, tps, thresholds)
 *   y_int      integer array n x m x H, NA already mapped to 0
 *   dims       c(n, m, H, C)
 *   iters      c(sample_iterations, burn_iterations, THIN, chain, SEED, constant_IRF, KERNEL(0/1), flags)
 *   hyper      c(theta_os, theta_ls, jitter)
 * returns list(theta, f, fstar, beta, threshold, loglik) as flat numeric vectors in the
 * layouts of gpirt_outputs; the R wrapper reshapes them. */
SEXP gpirt_run_call(SEXP y_int, SEXP dims, SEXP iters, SEXP hyper, SEXP theta_init, SEXP bpm,
                    SEXP bps, SEXP tpm, SEXP tps, SEXP thresholds) {
  gpirt_config cfg;
  gpirt_config_defaults(&cfg);
  const int *d = INTEGER(dims), *it = INTEGER(iters);
  const double *hy = REAL(hyper);
  cfg.n = d[0]; cfg.m = d[1]; cfg.H = d[2]; cfg.C = d[3];
  cfg.sample_iterations = it[0]; cfg.burn_iterations = it[1]; cfg.thin = it[2]; cfg.chain = it[3];
  cfg.seed = (uint64_t)it[4]; cfg.constant_irf = it[5]; cfg.kernel = it[6]; cfg.flags = it[7];
  cfg.theta_os = hy[0]; cfg.theta_ls = hy[1]; cfg.jitter = hy[2];
  cfg.progress = progress_cb;

  const R_xlen_t N = (R_xlen_t)cfg.n * cfg.m * cfg.H;
  int8_t *y = (int8_t *)R_alloc((size_t)N, 1);
  const int *yi = INTEGER(y_int);
  for (R_xlen_t k = 0; k < N; ++k) y[k] = (yi[k] == NA_INTEGER || yi[k] < 0) ? 0 : (int8_t)yi[k];

  const int P = cfg.constant_irf ? 1 : cfg.H;
  const int K = cfg.sample_iterations / (cfg.thin > 0 ? cfg.thin : 1);
  const int total = cfg.sample_iterations + cfg.burn_iterations;
  SEXP out = PROTECT(Rf_allocVector(VECSXP, 6));
  SEXP th = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)cfg.n * cfg.H * K));
  SEXP f = PROTECT(Rf_allocVector(REALSXP, (cfg.flags & GPIRT_RETURN_F) ? (R_xlen_t)cfg.n * cfg.m * cfg.H * K : 0));
  SEXP fs = PROTECT(Rf_allocVector(REALSXP, (cfg.flags & GPIRT_RETURN_FSTAR) ? (R_xlen_t)cfg.n_grid * cfg.m * P * K : 0));
  SEXP be = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)3 * cfg.m * P * K));
  SEXP tr = PROTECT(Rf_allocVector(REALSXP, (R_xlen_t)cfg.m * (cfg.C + 1) * P * K));
  SEXP ll = PROTECT(Rf_allocVector(REALSXP, total));
  gpirt_outputs o;
  memset(&o, 0, sizeof o);
  o.theta = REAL(th);
  o.f = XLENGTH(f) ? REAL(f) : NULL;
  o.fstar = XLENGTH(fs) ? REAL(fs) : NULL;
  o.beta = REAL(be); o.threshold = REAL(tr); o.loglik = REAL(ll);
  int rc = gpirt_run(&cfg, y, opt_real(theta_init), opt_real(bpm), opt_real(bps), opt_real(tpm),
                     opt_real(tps), opt_real(thresholds), &o);
  if (rc == GPIRT_ERR_INTERRUPT) { UNPROTECT(7); Rf_error("gpirt: interrupted by the user"); }
  if (rc != GPIRT_OK) { UNPROTECT(7); Rf_error("gpirt: %s (status %d)", gpirt_last_error(), rc); }
  SET_VECTOR_ELT(out, 0, th); SET_VECTOR_ELT(out, 1, f); SET_VECTOR_ELT(out, 2, fs);
  SET_VECTOR_ELT(out, 3, be); SET_VECTOR_ELT(out, 4, tr); SET_VECTOR_ELT(out, 5, ll);
  UNPROTECT(7);
  return out;
}
