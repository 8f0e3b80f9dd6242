// How fast can workgroups stream 128-row panels of a column-major n x n matrix, 16 columns
// (16 pieces of 1 KB, one leading dimension apart) per step with one step in flight -- the
// access pattern of the GEMM's A operand -- compared with the same bytes laid out contiguously
// per panel?  Prints the time and rate for several numbers of resident workgroups.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// panel bi covers columns [0, ncols); colstride / rowbase choose the layout
__global__ __launch_bounds__(256, 2) void stream(const double *__restrict__ A, int64_t ld, int npanels,
                                                 int ncols, int panel_major, double *out) {
  const int tid = threadIdx.x;
  double acc = 0.0;
  for (int bi = blockIdx.x; bi < npanels; bi += gridDim.x) {
    const double *base = panel_major ? A + (int64_t)bi * 128 * ld : A + (int64_t)bi * 128;
    const int64_t cs = panel_major ? 128 : ld;   // distance between consecutive columns of the panel
    double2 r[8];
#pragma unroll
    for (int p = 0; p < 8; ++p)
      r[p] = *reinterpret_cast<const double2 *>(base + ((tid & 63) << 1) + cs * (int64_t)((tid >> 6) + 4 * p));
    for (int k0 = 16; k0 < ncols; k0 += 16) {
      double2 q[8];
#pragma unroll
      for (int p = 0; p < 8; ++p)
        q[p] = *reinterpret_cast<const double2 *>(base + ((tid & 63) << 1) + cs * (int64_t)(k0 + (tid >> 6) + 4 * p));
#pragma unroll
      for (int p = 0; p < 8; ++p) acc += r[p].x + r[p].y;
      // ~ the matrix work of one slab when two workgroups share a CU
      for (int w = 0; w < 40; ++w) acc = fma(acc, 1.0000001, 1e-9);
#pragma unroll
      for (int p = 0; p < 8; ++p) r[p] = q[p];
    }
#pragma unroll
    for (int p = 0; p < 8; ++p) acc += r[p].x + r[p].y;
  }
  if (acc == 12345.678) out[0] = acc;
}

int main() {
  const int n = 10112, npanels = n / 128;
  double *A, *out;
  CK(hipMalloc(&A, sizeof(double) * (size_t)n * n));
  CK(hipMalloc(&out, 64));
  CK(hipMemset(A, 0, sizeof(double) * (size_t)n * n));
  hipEvent_t a, b;
  CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  for (int panel_major = 0; panel_major < 2; ++panel_major)
    for (int ncols : {n / 8, n}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(stream, dim3(npanels), dim3(256), 0, 0, A, (int64_t)n, npanels, ncols, panel_major, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
      }
      float ms;
      CK(hipEventElapsedTime(&ms, a, b));
      const double bytes = (double)npanels * 128 * ncols * 8;
      printf("%s layout, %5d columns per panel, %d workgroups: %.3f ms, %.2f TB/s, %.2f us per step\n",
             panel_major ? "panel-major " : "column-major", ncols, npanels, ms, bytes / ms * 1e-9,
             ms * 1e3 / (ncols / 16));
    }
  return 0;
}
