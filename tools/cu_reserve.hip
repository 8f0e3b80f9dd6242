// Probe: (1) which compute units do workgroups land on (HW_ID / XCC_ID hardware registers)?
// (2) if the workgroups of a chip-filling persistent kernel that find themselves on a few chosen
// CUs exit at once, do those CUs stay free, i.e. does a one-workgroup kernel that needs a whole
// CU's LDS (134 KB) start promptly on a second stream beside the flood?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <set>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ inline unsigned cu_key() {
  const unsigned hw = __builtin_amdgcn_s_getreg(4 | (31 << 11));     // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg(20 | (31 << 11)) & 0xF;   // HW_REG_XCC_ID
  const unsigned cu = (hw >> 8) & 0xF, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  return (xcc << 8) | (se << 5) | (sh << 4) | cu;
}

__global__ __launch_bounds__(256, 2) void whoami(unsigned *keys, unsigned *raw, int iters) {
  extern __shared__ double sh[];
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  if (threadIdx.x == 0) {
    keys[blockIdx.x] = cu_key();
    raw[blockIdx.x] = __builtin_amdgcn_s_getreg(4 | (31 << 11));
  }
  if (a == 12345.678) keys[0] = 0;
}

// persistent flood: workgroups on a reserved CU leave at once, the others share `ntiles` units of
// `iters` work through an atomic counter
__global__ __launch_bounds__(256, 2) void flood(const unsigned *reserved, int nres, int *counter, int ntiles,
                                                int iters, double *out, int *left) {
  extern __shared__ double sh[];
  __shared__ int tile;
  const unsigned key = cu_key();
  bool res = false;
  for (int i = 0; i < nres; ++i) res |= reserved[i] == key;
  if (res) { if (threadIdx.x == 0) atomicAdd(left, 1); return; }
  double a = threadIdx.x;
  for (;;) {
    if (threadIdx.x == 0) tile = atomicAdd(counter, 1);
    __syncthreads();
    const int t = tile;
    __syncthreads();
    if (t >= ntiles) break;
    for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  }
  if (a == 12345.678) out[0] = a;
}

__global__ __launch_bounds__(256) void small(double *out, int iters) {
  extern __shared__ double sh[];
  double a = threadIdx.x;
  for (int i = 0; i < iters; ++i) { a = a * 1.0000001 + 0.5; sh[threadIdx.x] = a; }
  if (a == 12345.678) out[0] = a;
}

int main() {
  double *out; unsigned *keys, *raw, *dres; int *counter, *left;
  CK(hipMalloc(&out, 64)); CK(hipMalloc(&keys, 4 * 8192)); CK(hipMalloc(&raw, 4 * 8192));
  CK(hipMalloc(&dres, 4 * 64)); CK(hipMalloc(&counter, 8)); left = counter + 1;
  CK(hipFuncSetAttribute((const void *)whoami, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  CK(hipFuncSetAttribute((const void *)flood, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));
  CK(hipFuncSetAttribute((const void *)small, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // (1) where do 2048 workgroups of 74 KB land?
  hipLaunchKernelGGL(whoami, dim3(2048), dim3(256), 74 * 1024, 0, keys, raw, 4000);
  CK(hipDeviceSynchronize());
  std::vector<unsigned> hk(2048), hr(2048);
  CK(hipMemcpy(hk.data(), keys, 4 * 2048, hipMemcpyDeviceToHost));
  CK(hipMemcpy(hr.data(), raw, 4 * 2048, hipMemcpyDeviceToHost));
  std::set<unsigned> distinct(hk.begin(), hk.end());
  printf("distinct CU keys: %zu\n", distinct.size());
  printf("first 16 workgroups: ");
  for (int i = 0; i < 16; ++i) printf("%03x ", hk[i]);
  printf("\nraw HW_ID of wg0: %08x\n", hr[0]);
  std::set<unsigned> xccs, ses, cus;
  for (unsigned k : distinct) { xccs.insert(k >> 8); ses.insert((k >> 5) & 7); cus.insert(k & 0x1F); }
  printf("xcc values %zu, se values %zu, (sh,cu) values %zu\n", xccs.size(), ses.size(), cus.size());
  // reserve one CU per XCD: the smallest key of each xcc
  std::vector<unsigned> res;
  for (unsigned x : xccs) for (unsigned k : distinct) if ((k >> 8) == x) { res.push_back(k); break; }
  printf("reserved:"); for (unsigned k : res) printf(" %03x", k); printf("\n");
  CK(hipMemcpy(dres, res.data(), 4 * res.size(), hipMemcpyHostToDevice));
  hipStream_t s1, s2;
  CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  for (int nres : {0, (int)res.size()}) {
    for (int lds_kb : {80, 134}) {
      CK(hipMemset(counter, 0, 8));
      CK(hipDeviceSynchronize());
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(small, dim3(1), dim3(256), lds_kb * 1024, s2, out, 4000);
      CK(hipStreamSynchronize(s2));
      double alone = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      // 4096 units of ~100 us over ~512 slots: ~0.8 ms of flood
      hipLaunchKernelGGL(flood, dim3(512 + 4 * 16), dim3(256), 74 * 1024, s1, dres, nres, counter, 4096, 20000, out, left);
      std::this_thread::sleep_for(std::chrono::microseconds(200));
      double beside[3];
      for (int rep = 0; rep < 3; ++rep) {
        t0 = std::chrono::steady_clock::now();
        hipLaunchKernelGGL(small, dim3(1), dim3(256), lds_kb * 1024, s2, out, 4000);
        CK(hipStreamSynchronize(s2));
        beside[rep] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      }
      t0 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s1));
      double rest = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      int h[2];
      CK(hipMemcpy(h, counter, 8, hipMemcpyDeviceToHost));
      printf("reserved CUs %d, small kernel LDS %3d KB: alone %.0f us; beside the flood %.0f %.0f %.0f us; flood ran %.0f us longer; "
             "%d workgroups left reserved CUs\n", nres, lds_kb, alone, beside[0], beside[1], beside[2], rest, h[1]);
    }
  }
  return 0;
}
