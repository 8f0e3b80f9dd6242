// Branch-free ordinal-probit cell for the device.
//
// log( Phi(zhi) - Phi(zlo) )  (log_like_fn.m:3-4 with the probit link of sigmoid_fn.m:4,
// ordinal form gpirt-sythetic.R:159-167) from two evaluations of the upper tail
//   Phi(-z) = exp(-z^2/2) * g(z),  z >= 0,
// where g is read from a table of degree-5 polynomials (phi_table.h, relative error <= 5e-16 on
// [0,40]) whose interval number and local variable are cut out of the bits of z + 1: no
// division (v_rcp_f64 alone costs as much as 3.4 fused multiply-adds, its Newton steps 4 more).
// No data-dependent branch: the library erfc picks one of several code paths per lane, and a
// wave whose lanes straddle them pays for all.
#pragma once
#include "common.h"

#define GPIRT_PHI_QUAL static __device__ const
#include "phi_table.h"

namespace gpirt {

constexpr int PHI_ROWS = GPIRT_PHI_K;
constexpr int PHI_NC = 6;        // doubles per row: c0..c5
constexpr int PHI_LD = PHI_NC;   // LDS row stride: 48-byte rows, three 16-byte reads, 16 bank positions
constexpr int PHI_S = GPIRT_PHI_S;

constexpr int EXP_TAB = 256;  // 2^(j/256), j = 0..255, behind the polynomial table
constexpr int PHI_LDS_DOUBLES = PHI_ROWS * PHI_LD + EXP_TAB;

// every thread of the workgroup calls this once, followed by __syncthreads()
__device__ inline void phi_table_to_lds(double *s) {
  for (int e = threadIdx.x; e < PHI_ROWS * PHI_NC; e += blockDim.x)
    s[(e / PHI_NC) * PHI_LD + (e % PHI_NC)] = kPhiTab[e];
  for (int j = threadIdx.x; j < EXP_TAB; j += blockDim.x)
    s[PHI_ROWS * PHI_LD + j] = exp2((double)j * (1.0 / EXP_TAB));
}

// exp(-z^2/2) for z >= 0: with z^2 = -(256 m + j) 2 ln2/256 + v, 0 <= j < 256, |v| <= ln2/256, 2^(j/256) comes from
// the table, exp(-v/2) - 1 from a degree-4 polynomial in v (remainder (v/2)^5/120 < 4e-17), scaled by 2^m (m <= 0).  v is formed
// from the EXACT square: n 2 ln2_hi/256 is exact (a 33-bit constant times an integer below 2^19), and one fused
// multiply-add takes it from z z before that square is ever rounded -- what rounding z^2/2 first would lose
// (2^-53 of it in the exponent, a relative error of the result that grows with z^2) never arises.  The signs
// and the factor 1/2 sit in the constants.  Branch-free; underflows gradually to zero.
__device__ inline double exp_neg_half_square(double z, const double *s) {
  const double nf = rint(z * z * 184.6649652337873);             // 128 / ln 2; z <= 40: fits an int with room
  double v = fma(z, z, -(nf * 0.0054152123475432746));           // 2 ln2/256 high part (33 bits: exact product)
  v = fma(nf, -5.812982117197185e-13, v);                        // low part
  const int n = (int)-nf;                                        // -nf = 256 m + j, 0 <= j < 256 (floor division)
  const double t = s[PHI_ROWS * PHI_LD + (n & (EXP_TAB - 1))];   // 2^(j/256)
  // (exp(-v/2) - 1) / v = -1/2 + v/8 - v^2/48 + v^3/384
  double q = fma(v, 1.0 / 384.0, -1.0 / 48.0);
  q = fma(q, v, 0.125);
  q = fma(q, v, -0.5);
  return ldexp(fma(t, q * v, t), n >> 8);
}

__device__ inline double phi_upper(double z, const double *s) {   // Phi(-z), z >= 0
  z = fmin(z, 40.0);                          // Phi(-40) is far below the smallest double
  const double w = z + 1.0;                   // in [1, 41]: exponent 0..5
  const int hi = __double2hiint(w);
  const int k = (hi >> (20 - PHI_S)) - (1023 << PHI_S);          // binade * 64 + top mantissa bits
  // lower end of the interval: the same high word with the lower mantissa bits cleared
  const double t = w - __hiloint2double(hi & ~((1 << (20 - PHI_S)) - 1), 0);
  int row = k * (PHI_LD * 8);
#ifdef GPIRT_PROBE_SAME_ROW
  row = 100 * (PHI_LD * 8);
#endif
  asm volatile("" : "+v"(row));               // one address, three immediate offsets
  // 16-byte LDS reads move 256 bytes per clock, the 8-byte pairs (ds_read2_b64) half of that
  const double2 *c = reinterpret_cast<const double2 *>(reinterpret_cast<const char *>(s) + row);
  const double2 c01 = c[0], c23 = c[1], c45 = c[2];
  double g = fma(c45.y, t, c45.x);
  g = fma(g, t, c23.y);
  g = fma(g, t, c23.x);
  g = fma(g, t, c01.y);
  g = fma(g, t, c01.x);
  return exp_neg_half_square(z, s) * g;
}

// P(zlo < Z <= zhi) from the two tails e1 = Phi(-|zlo|), e2 = Phi(-|zhi|) (zlo, zhi already
// clamped to finite values), floored so that a vanished cell stays finite
__device__ inline double ordinal_cell_from_tails(double zlo, double zhi, double e1, double e2) {
  // both thresholds on one side of zero: e1 - e2 (zlo >= 0) or e2 - e1 (zhi <= 0), and whichever applies is
  // the positive one of the two (the tail decreases in |z|), i.e. |e1 - e2| exactly; thresholds on both sides:
  // (1 - e1) - e2.  As two sums and one select: written as a chain of conditionals it compiled to divergent
  // branches in the middle of the row loops.
  const double same_side = fabs(e1 - e2), across = (1.0 - e1) - e2;
  double p = (zlo < 0.0 && zhi > 0.0) ? across : same_side;
  if (!(p > kPFloor)) p = kPFloor;
  return p;
}

__device__ inline double ordinal_cell_prob(double zlo, double zhi, const double *s) {
  // (an end category's threshold is -+inf: phi_upper cuts |z| at 40, the comparisons above take infinities)
  return ordinal_cell_from_tails(zlo, zhi, phi_upper(fabs(zlo), s), phi_upper(fabs(zhi), s));
}

__device__ inline double ordinal_cell_fast(double zlo, double zhi, const double *s) {
  return log_pos(ordinal_cell_prob(zlo, zhi, s), kLogTab);
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
  __device__ inline double log_value() const { return log_pos(m, kLogTab) + (double)e * 0.69314718055994530942; }
};

}  // namespace gpirt
