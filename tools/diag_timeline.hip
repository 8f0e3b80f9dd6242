// Where the diagonal-block kernel of the Cholesky spends its clocks: potrf_diag.hip built with
// DIAG_STAMPS records the shader clock of thread 0 at every phase boundary.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -DDIAG_STAMPS -I gpirt_amd/csrc -I include tools/diag_timeline.hip -o tools/diag_timeline
#include "../gpirt_amd/csrc/potrf_diag.hip"
#include <cstdio>
#include <vector>
#include <cmath>
namespace gpirt { void set_error(const char *, ...) {} }
int main() {
  const int n = 128;
  std::vector<double> x(n), S((size_t)n * n);
  for (int i = 0; i < n; ++i) x[i] = sin(1.0 + i) * 2.0;
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) S[i + (size_t)n * j] = exp(-0.5 * (x[i] - x[j]) * (x[i] - x[j])) + (i == j ? 1e-6 : 0.0);
  double *dS, *dI; int *dinfo;
  hipMalloc(&dS, sizeof(double) * n * n); hipMalloc(&dI, sizeof(double) * 512 * 512); hipMalloc(&dinfo, 4);
  hipMemset(dI, 0, sizeof(double) * 512 * 512); hipMemset(dinfo, 0, 4);
  const char *names[19] = {"start", "block in", "p0 factor", "p0 panel", "p0 trailing", "p1 factor", "p1 panel", "p1 trailing",
                           "p2 factor", "p2 panel", "p2 trailing", "p3 factor", "p3 panel", "p3 trailing", "L out", "inv level 1",
                           "inv level 2a", "inv level 2b", "Inv out"};
  for (int rep = 0; rep < 3; ++rep) {
    hipMemcpy(dS, S.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
    gpirt::launch_potrf_diag(0, dS, n, 0, 1, dI, 1, dinfo, false);
    hipDeviceSynchronize();
  }
  long long st[32];
  hipMemcpyFromSymbol(st, HIP_SYMBOL(gpirt::g_diag_stamps), sizeof(st));
  for (int i = 1; i < 19; ++i) if (st[i] && st[i - 1]) printf("%-14s %8lld clocks\n", names[i], st[i] - st[i - 1]);
  printf("total %lld clocks\n", st[18] - st[0]);
  printf("from the first barrier: wave 1 begins its columns %lld, first batch done %lld, second %lld; wave 0 ends its factorisation %lld\n",
         st[20] - st[1], st[21] - st[1], st[22] - st[1], st[23] - st[1]);
  return 0;
}
