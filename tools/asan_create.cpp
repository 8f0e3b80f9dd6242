// Host-side AddressSanitizer run of the sampler's set-up and one sweep at ragged sizes (device
// code is built as usual; only the host half carries the sanitizer).  Build and run:
//   hipcc -O1 -g -std=c++17 --offload-arch=gfx950 -Xarch_host -fsanitize=address \
//         -Xarch_host -fno-omit-frame-pointer gpirt_amd/csrc/*.hip tools/asan_create.cpp -o tools/asan_create
//   ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 tools/asan_create
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <unistd.h>

#include "../include/gpirt.h"

static int run(int n, int m, int H, int constant_irf, int flags) {
  gpirt_config cfg;
  gpirt_config_defaults(&cfg);
  cfg.n = n; cfg.m = m; cfg.H = H; cfg.C = 3; cfg.seed = 9; cfg.theta_ls = H > 2 ? H / 2.0 : 1.0;
  cfg.constant_irf = constant_irf; cfg.flags = flags;
  std::vector<int8_t> y((size_t)n * m * H);
  std::vector<double> theta((size_t)n * H);
  unsigned s = 12345u + n + 31 * m + 977 * H;
  for (auto &v : y) { s = s * 1664525u + 1013904223u; v = (int8_t)((s >> 24) & 3); }
  for (auto &v : theta) { s = s * 1664525u + 1013904223u; v = ((s >> 8) / 16777216.0 - 0.5) * 4.0; }
  gpirt_sampler *sm = nullptr;
  int rc = gpirt_sampler_create(&cfg, y.data(), theta.data(), nullptr, nullptr, nullptr, nullptr, nullptr, &sm);
  if (rc) { printf("create n=%d m=%d H=%d: rc=%d %s\n", n, m, H, rc, gpirt_last_error()); return rc; }
  rc = gpirt_sampler_sweep(sm, 2);
  const int P = constant_irf ? 1 : H;
  std::vector<double> th((size_t)n * H), f((size_t)n * m * H), fs((size_t)cfg.n_grid * m * P), be((size_t)3 * m * P),
      tr((size_t)m * (cfg.C + 1) * P);
  if (!rc) rc = gpirt_sampler_get_state(sm, th.data(), f.data(), fs.data(), be.data(), tr.data());
  std::vector<char> blob((size_t)gpirt_sampler_checkpoint_bytes(sm));
  if (!rc) rc = gpirt_sampler_checkpoint(sm, blob.data(), (int64_t)blob.size());
  if (!rc) rc = gpirt_sampler_restore(sm, blob.data(), (int64_t)blob.size());
  if (!rc) rc = gpirt_sampler_accumulate(sm, 0);
  gpirt_sampler_destroy(sm);
  printf("n=%d m=%d H=%d const=%d flags=%d: rc=%d theta[0]=%g\n", n, m, H, constant_irf, flags, rc, th[0]);
  return rc;
}

int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const int cases[][5] = {{1, 1, 1, 0, 0}, {5, 3, 2, 0, 0}, {100, 10, 10, 0, 0}, {129, 2, 1, 0, 0}, {513, 3, 1, 0, 0},
                          {7, 2, 40, 1, 0}, {100, 4, 6, 1, 0}, {100, 9, 1, 0, GPIRT_FSTAR_SOLVE},
                          {100, 9, 1, 0, GPIRT_EXACT_FSTAR}, {300, 5, 1, 0, GPIRT_GENERATOR_PRIOR},
                          {1100, 3, 1, 0, 0}, {2100, 130, 1, 0, 0}, {60, 1030, 1, 0, 0}};
  int bad = 0;
  for (const auto &c : cases) bad |= run(c[0], c[1], c[2], c[3], c[4]) != 0;
  printf(bad ? "FAILED\n" : "all sizes passed under the host sanitizer\n");
  // the HIP runtime's own teardown trips a check of the sanitizer's device allocator: skip the
  // finalizers, every buffer of the cases above has been freed and checked by now
  _exit(bad);
}
