// Diagonal-block kernel of the blocked Cholesky (chol inside mvnrnd, utils/sample_fs.m:2):
// factor one 128x128 block and invert the factor.  Kept in its own file because the fully
// unrolled code takes long to compile.
#include "common.h"

namespace gpirt {

// ---------------------------------------------------------------------------------------
// Diagonal-block factorisation: one workgroup factors a 128x128 block and inverts the
// factor.  The block lives in registers: thread (r, h) = (tid/2, tid%2) owns the 64 entries
// of row r in the column pairs {4q+2h, 4q+2h+1}, q = 0..31, so both threads of a row carry
// the same load at every step and read the pivot column from LDS as 16-byte pairs.  The
// 128 pivot steps are fully unrolled (static register indices); each costs one barrier.
// The inverse X = L^-1 is then built row by row with no barrier at all: thread (t, h) owns
// the same column pairs of column t of X and the two halves meet in one lane shuffle.
// ---------------------------------------------------------------------------------------
constexpr int XLD = TILE + 2;   // row stride of the LDS image: column pairs stay 16-byte aligned

// The block stays in LDS (row-major, stride XLD).  Thread (r, h) = (tid/2, tid%2) owns, in row
// r, the column pairs 2p, 2p+1 with p % 2 == h.  A pivot step costs ONE barrier: right after
// updating its pairs a thread copies its (already final, still unscaled) entry of the next
// pivot column into a double-buffered column vector; after the barrier every thread reads the
// pivot d and that vector and applies T[r][c] -= (u_r / d) * u_c on the lower triangle.
__global__ __launch_bounds__(256) void potrf_diag_kernel(double *__restrict__ S, int n_pad, int kb,
                                                        double *__restrict__ inv_diag, int nb,
                                                        int *__restrict__ info) {
  __shared__ __attribute__((aligned(16))) double T[TILE * XLD];
  __shared__ __attribute__((aligned(16))) double colb[2][TILE];
  const int z = blockIdx.x, tid = threadIdx.x, r = tid >> 1, h = tid & 1;
  double *A = S + (int64_t)n_pad * n_pad * z + (int64_t)kb * TILE * (n_pad + 1);
  double *Inv = inv_diag + (int64_t)TILE * TILE * (kb + (int64_t)nb * z);

  for (int e = tid; e < TILE * TILE; e += 256) {
    const int i = e & (TILE - 1), c = e >> 7;
    T[i * XLD + c] = (i >= c) ? A[i + (int64_t)n_pad * c] : 0.0;
  }
  __syncthreads();
  double *Tr = &T[r * XLD];
  if (h == 0) colb[0][r] = Tr[0];
  __syncthreads();

  for (int j = 0; j < TILE; ++j) {
    const double *cb = colb[j & 1];
    double d = cb[j];
    if (!(d > 0.0)) {
      if (tid == 0) atomicMax(info, kb * TILE + j + 1);
      d = 1.0;
    }
    const double rs = 1.0 / sqrt(d);
    const double ur = cb[r];
    const double w = ur * (rs * rs);
    const int pj = j >> 1;                  // pair holding the pivot column
    if (r >= j && (pj & 1) == h) {
      Tr[j] = ur * rs;                      // L[r][j]
      if (!(j & 1) && r > j) Tr[j + 1] -= w * cb[j + 1];
    }
    if (r > j) {
      // first pair strictly right of the pivot pair that this thread owns, up to the diagonal
      int p = pj + 1;
      if ((p & 1) != h) ++p;
      const int pend = r >> 1;
      for (; p <= pend; p += 2) {
        const double2 cc = *reinterpret_cast<const double2 *>(&cb[2 * p]);
        double2 v = *reinterpret_cast<double2 *>(&Tr[2 * p]);
        v.x -= w * cc.x;
        v.y -= w * cc.y;
        *reinterpret_cast<double2 *>(&Tr[2 * p]) = v;
      }
    }
    // publish this row's entry of the next pivot column (final after the update above)
    if (j + 1 < TILE && (((j + 1) >> 1) & 1) == h) colb[(j + 1) & 1][r] = Tr[j + 1];
    __syncthreads();
  }

  // clear what the update loop wrote above the diagonal (entry (r, r+1) for even r), then
  // L back to global
  if (h == 0 && !(r & 1)) Tr[r + 1] = 0.0;
  __syncthreads();
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int i = e & (TILE - 1), c = e >> 7;
    A[i + (int64_t)n_pad * c] = T[i * XLD + c];
  }

  // X = L^-1, column t by the thread pair (t, h): row i of X needs sum_{c<i} L[i][c] X[c][t];
  // the two threads split the columns c by pair parity and meet in one lane shuffle.  X is
  // written into the strict upper triangle of T (X[i][t] at T[t][i], i > t) and its diagonal
  // into colb[0], so no second LDS image is needed.
  const int t = r;
  double *Xt = &T[t * XLD];
  const double dt = 1.0 / Tr[t];
  __syncthreads();
  if (h == 0) colb[0][t] = dt;
  for (int i = t + 1; i < TILE; ++i) {
    const double *Li = &T[i * XLD];
    double s = 0.0, s2 = 0.0;
    // c = t: X[t][t] = dt (not yet in the upper triangle)
    if (((t >> 1) & 1) == h) s = Li[t] * dt;
    int p = (t + 1) >> 1;                   // pair holding column t+1
    if ((p & 1) != h) ++p;
    for (; 2 * p < i; p += 2) {
      const int c = 2 * p;
      const double2 l = *reinterpret_cast<const double2 *>(&Li[c]);
      const double2 xx = *reinterpret_cast<const double2 *>(&Xt[c]);
      if (c > t) s += l.x * xx.x;
      if (c + 1 < i) s2 += l.y * xx.y;
    }
    s += s2;
    s += __shfl_xor(s, 1, 64);
    const double val = -s / Li[i];
    if (h == 0) Xt[i] = val;
    // the partner reads Xt[i] in later rows: same wave, LDS ops execute in order
  }
  __syncthreads();
  for (int e = tid; e < TILE * TILE; e += 256) {
    const int i = e & (TILE - 1), c = e >> 7;
    Inv[i + TILE * c] = (i > c) ? T[c * XLD + i] : (i == c ? colb[0][i] : 0.0);
  }
}

int launch_potrf_diag(hipStream_t st, double *S, int n_pad, int kb, int batch, double *inv_diag, int nb,
                      int *d_info) {
  hipLaunchKernelGGL(potrf_diag_kernel, dim3(batch), dim3(256), 0, st, S, n_pad, kb, inv_diag, nb,
                     d_info);
  GP_HIP(hipGetLastError());
  return GPIRT_OK;
}

}  // namespace gpirt
