// Do streams created with a compute-unit mask keep a large grid off a reserved set of CUs,
// so that a chain of small kernels on a second stream (masked to the reserved CUs) starts at
// once beside it?  Prints (1) how many distinct CUs each stream's workgroups ran on, (2) the
// duration of a chain of 20 one-workgroup kernels (134 KB LDS) and of 20 256-workgroup kernels,
// alone and beside the large grid.
//
// Result on the MI355X boxes used for this work (ROCm 7.2): hipExtStreamCreateWithCUMask itself
// ends the process with a segmentation fault in the runtime (host side, before any kernel runs),
// so a per-stream CU reservation is not available there.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned where_am_i() {
  unsigned hw = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_ID, 32 bits
  unsigned xcc = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11));   // XCC_ID, 4 bits
  unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  return (xcc << 12) | (se << 8) | (sh << 4) | cu;
}

__global__ __launch_bounds__(256, 2) void big(double *out, int iters, unsigned *where) {
  extern __shared__ double sh[];
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  if (a == 12345.678) out[0] = a;
  if (where && threadIdx.x == 0) where[blockIdx.x] = where_am_i();
}

__global__ __launch_bounds__(256) void small(double *out, int iters, unsigned *where) {
  extern __shared__ double sh[];
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  if (a == 12345.678) out[0] = a;
  if (where && threadIdx.x == 0) where[blockIdx.x] = where_am_i();
}

static int distinct(unsigned *d_where, int count) {
  std::vector<unsigned> h(count);
  CK(hipMemcpy(h.data(), d_where, count * sizeof(unsigned), hipMemcpyDeviceToHost));
  std::set<unsigned> s(h.begin(), h.end());
  std::set<unsigned> xcc;
  for (unsigned v : s) xcc.insert(v >> 12);
  printf("  %d workgroups on %zu distinct CUs in %zu XCDs\n", count, s.size(), xcc.size());
  return (int)s.size();
}

int main(int argc, char **argv) {
  const int reserve_per_word = argc > 1 ? atoi(argv[1]) : 4;   // reserved bits in each 32-bit mask word
  setvbuf(stdout, nullptr, _IONBF, 0);
  double *out;
  unsigned *where;
  CK(hipMalloc(&out, 64));
  CK(hipMalloc(&where, 8192 * sizeof(unsigned)));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int ncu = prop.multiProcessorCount, words = (ncu + 31) / 32;
  printf("%d CUs, %d mask words, reserving %d bits per word\n", ncu, words, reserve_per_word);
  std::vector<uint32_t> bulk_mask(words), chain_mask(words);
  for (int w = 0; w < words; ++w) {
    uint32_t r = reserve_per_word >= 32 ? 0xffffffffu : ((1u << reserve_per_word) - 1u);
    chain_mask[w] = r;
    bulk_mask[w] = ~r;
  }
  hipStream_t sb, sc, plain;
  printf("creating masked streams\n");
  CK(hipExtStreamCreateWithCUMask(&sb, words, bulk_mask.data()));
  CK(hipExtStreamCreateWithCUMask(&sc, words, chain_mask.data()));
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  printf("streams created\n");
  CK(hipFuncSetAttribute((const void *)big, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));

  printf("bulk stream:\n");
  hipLaunchKernelGGL(big, dim3(4096), dim3(256), 74 * 1024, sb, out, 2000, where);
  CK(hipDeviceSynchronize());
  distinct(where, 4096);
  printf("chain stream:\n");
  hipLaunchKernelGGL(small, dim3(4096), dim3(256), 8 * 1024, sc, out, 2000, where);
  CK(hipDeviceSynchronize());
  distinct(where, 4096);
  printf("unmasked stream:\n");
  hipLaunchKernelGGL(small, dim3(4096), dim3(256), 8 * 1024, plain, out, 2000, where);
  CK(hipDeviceSynchronize());
  distinct(where, 4096);

  auto chain = [&](hipStream_t s, int wgs, int lds_kb, int iters) {
    auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < 20; ++i)
      hipLaunchKernelGGL(small, dim3(wgs), dim3(256), lds_kb * 1024, s, out, iters, (unsigned *)nullptr);
    CK(hipStreamSynchronize(s));
    return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  };
  for (int variant = 0; variant < 2; ++variant) {
    const int wgs = variant ? 256 : 1, lds = variant ? 40 : 134, iters = variant ? 1000 : 4000;
    for (hipStream_t s : {sc, plain}) {
      chain(s, wgs, lds, iters);   // warm
      double alone = chain(s, wgs, lds, iters);
      hipLaunchKernelGGL(big, dim3(512 * 16), dim3(256), 74 * 1024, sb, out, 20000, (unsigned *)nullptr);
      std::this_thread::sleep_for(std::chrono::microseconds(300));
      double beside = chain(s, wgs, lds, iters);
      auto t0 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(sb));
      double rest = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      printf("chain of 20 x %3d workgroups (%3d KB LDS) on the %s stream: alone %.0f us, beside the masked "
             "bulk grid %.0f us (bulk ran %.0f us longer)\n",
             wgs, lds, s == sc ? "reserved-CU" : "unmasked", alone, beside, rest);
    }
  }
  return 0;
}
