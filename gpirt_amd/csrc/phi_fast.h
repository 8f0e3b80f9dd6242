// Branch-free ordinal-probit cell for the device.
//
// log( Phi(zhi) - Phi(zlo) )  (log_like_fn.m:3-4 with the probit link of sigmoid_fn.m:4,
// ordinal form gpirt-sythetic.R:159-167) from two evaluations of the upper tail
//   Phi(-z) = exp(-z^2/2) * g(z),  z >= 0,
// where g is read from a 64-interval table of degree-6 polynomials (phi_table.h, relative
// error <= 5e-16 on [0,40]).  No data-dependent branch: the library erfc picks one of
// several code paths per lane, and a wave whose lanes straddle them pays for all.
#pragma once
#include "common.h"

#define GPIRT_PHI_QUAL static __device__ const
#include "phi_table.h"

namespace gpirt {

constexpr int PHI_ROWS = GPIRT_PHI_K;
constexpr int PHI_NC = GPIRT_PHI_DEG + 1;
constexpr int PHI_LD = 9;   // LDS row stride (doubles): 72-byte rows spread the gathers over banks

constexpr int EXP_TAB = 64;  // 2^(j/64), j = 0..63, behind the polynomial table
constexpr int PHI_LDS_DOUBLES = PHI_ROWS * PHI_LD + EXP_TAB;

// every thread of the workgroup calls this once, followed by __syncthreads()
__device__ inline void phi_table_to_lds(double *s) {
  for (int e = threadIdx.x; e < PHI_ROWS * PHI_NC; e += blockDim.x)
    s[(e / PHI_NC) * PHI_LD + (e % PHI_NC)] = kPhiTab[e];
  for (int j = threadIdx.x; j < EXP_TAB; j += blockDim.x)
    s[PHI_ROWS * PHI_LD + j] = exp2((double)j * (1.0 / EXP_TAB));
}

// exp(h + l) for h <= 0 and a tiny correction l: h = (64 m + j) ln2/64 + r with |r| <= ln2/128,
// 2^(j/64) from the table, exp(r) - 1 from a degree-5 polynomial (remainder 3e-17), scaled
// by 2^m.  Branch-free; underflows gradually to zero.
__device__ inline double exp_neg(double h, double l, const double *s) {
  const double hc = fmax(h, -745.5);
  const double nf = rint(hc * 92.33248261689366);              // 64 / ln 2
  double r = fma(nf, -0.010830424695086549, hc);                // ln2/64 high part (33 bits: exact product)
  r = fma(nf, -1.162596423439437e-12, r) + l;                   // low part
  const int n = (int)nf;
  const double t = s[PHI_ROWS * PHI_LD + (n & (EXP_TAB - 1))];
  double q = fma(r, 1.0 / 120.0, 1.0 / 24.0);
  q = fma(q, r, 1.0 / 6.0);
  q = fma(q, r, 0.5);
  q = fma(q, r, 1.0);
  return ldexp(fma(t, q * r, t), n >> 6);
}

__device__ inline double phi_upper(double z, const double *s) {   // Phi(-z), z >= 0
  z = fmin(z, 40.0);                          // Phi(-40) is far below the smallest double
  const double d = 1.0 + 0.5 * z;
  double u = __builtin_amdgcn_rcp(d);
  u = fma(u, fma(-d, u, 1.0), u);
  u = fma(u, fma(-d, u, 1.0), u);
  const double sc = u * (double)PHI_ROWS;
  const int k = min((int)sc, PHI_ROWS - 1);
  const double t = 2.0 * (sc - (double)k) - 1.0;
  const double *c = s + k * PHI_LD;
  double g = c[6];
  g = fma(g, t, c[5]);
  g = fma(g, t, c[4]);
  g = fma(g, t, c[3]);
  g = fma(g, t, c[2]);
  g = fma(g, t, c[1]);
  g = fma(g, t, c[0]);
  const double h = -0.5 * z * z;
  const double l = fma(-0.5 * z, z, -h);      // rounding error of h, folded into the exponent
  return exp_neg(h, l, s) * g;
}

// P(zlo < Z <= zhi) from the two tails e1 = Phi(-|zlo|), e2 = Phi(-|zhi|) (zlo, zhi already
// clamped to finite values), floored so that a vanished cell stays finite
__device__ inline double ordinal_cell_from_tails(double zlo, double zhi, double e1, double e2) {
  double p = zlo >= 0.0 ? e1 - e2 : (zhi <= 0.0 ? e2 - e1 : (1.0 - e1) - e2);
  if (!(p > kPFloor)) p = kPFloor;
  return p;
}

__device__ inline double ordinal_cell_prob(double zlo, double zhi, const double *s) {
  zlo = fmax(zlo, -1e30);                      // +-inf thresholds: keep the arithmetic finite
  zhi = fmin(zhi, 1e30);
  return ordinal_cell_from_tails(zlo, zhi, phi_upper(fabs(zlo), s), phi_upper(fabs(zhi), s));
}

__device__ inline double ordinal_cell_fast(double zlo, double zhi, const double *s) {
  return log(ordinal_cell_prob(zlo, zhi, s));
}

// Sum of log-probabilities as the log of their product: the running product is kept as
// mantissa x 2^exponent (renormalised at every factor, so it never underflows), which costs
// four instructions per cell where a log costs about thirty; one log per lane at the end.
// Each factor adds one rounding of relative size 2^-53 to the product, i.e. the same
// absolute error to the sum of logs -- no worse than adding rounded logs.
struct LogProduct {
  double m;
  int e;
  __device__ inline void reset() { m = 1.0; e = 0; }
  __device__ inline void mul(double p) {       // p in [kPFloor, 1]
    const double q = m * p;
    m = __builtin_amdgcn_frexp_mant(q);
    e += __builtin_amdgcn_frexp_exp(q);
  }
  __device__ inline double log_value() const { return log(m) + (double)e * 0.69314718055994530942; }
};

}  // namespace gpirt
