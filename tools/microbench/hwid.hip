// Which CU does the dispatcher put workgroups on?  (XCC_ID / HW_ID registers.)  Usage: hwid [n_workgroups] [lds_bytes] [threads]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
__global__ void k(unsigned* out, unsigned* cnt, unsigned* maxc) {
  unsigned cu = 0;
  if (threadIdx.x == 0) {
    unsigned hw = __builtin_amdgcn_s_getreg(((8 - 1) << 11) | (8 << 6) | 4);     // HW_ID[15:8]: cu_id, sh_id, se_id
    unsigned xcc = __builtin_amdgcn_s_getreg(((4 - 1) << 11) | (0 << 6) | 20);   // XCC_ID[3:0]
    cu = (xcc << 8) | hw;
    out[blockIdx.x] = cu;
    atomicMax(maxc + cu, atomicAdd(cnt + cu, 1u) + 1u);                            // workgroups resident on this CU right now
  }
  long long t0 = clock64(); while (clock64() - t0 < 20000000) {}   // stay resident ~10 ms
  if (threadIdx.x == 0) atomicSub(cnt + cu, 1u);
}
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256, lds = argc > 2 ? atoi(argv[2]) : 35400, thr = argc > 3 ? atoi(argv[3]) : 128;
  unsigned* d; (void)hipMalloc(&d, n * 4);
  unsigned* tab; (void)hipMalloc(&tab, 2 * 4096 * 4); (void)hipMemset(tab, 0, 2 * 4096 * 4);
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
  hipLaunchKernelGGL(k, dim3(n), dim3(thr), lds, 0, d, tab, tab + 4096);
  std::vector<unsigned> h(n); (void)hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
  std::map<unsigned, int> per; for (int i = 0; i < n; i++) per[h[i]]++;
  int hist[16] = {0}; for (auto& kv : per) hist[kv.second < 15 ? kv.second : 15]++;
  printf("%d workgroups (%d B LDS, %d threads): %zu CUs used; CUs hosting c workgroups:", n, lds, thr, per.size());
  for (int i = 1; i < 16; i++) if (hist[i]) printf(" c=%d:%d", i, hist[i]);
  std::vector<unsigned> mc(4096); (void)hipMemcpy(mc.data(), tab + 4096, 4096 * 4, hipMemcpyDeviceToHost);
  int mh[16] = {0}; for (int i = 0; i < 4096; i++) if (mc[i]) mh[mc[i] < 15 ? mc[i] : 15]++;
  printf("; CUs with at most r resident at once:");
  for (int i = 1; i < 16; i++) if (mh[i]) printf(" r=%d:%d", i, mh[i]);
  printf("\n");
  return 0;
}
