// Latency microbenchmarks for a single wavefront on gfx950: dependent VALU chain, dependent LDS pointer chase,
// taken-branch loop, 64-bit shifts, dependent global (L2) loads.  Prints cycles per operation (s_memtime based).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_valu(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t v = seed + threadIdx.x;
  uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 1024; i++) v = v * 3u + 1u;   // v_mad / lshl_add chain
  uint64_t t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_add(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t v = seed + threadIdx.x, w = seed;
  uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 1024; i++) { v = (v ^ w) + 0x9e37u; }   // 2 dependent simple ops
  uint64_t t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_shift64(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint64_t v = ((uint64_t)seed << 32) | threadIdx.x; uint32_t s = seed & 7;
  uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll
  for (int i = 0; i < 1024; i++) { v = (v >> (s + (uint32_t)(v & 3))) | 0x8000000000000000ull; }
  uint64_t t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = (uint32_t)v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_lds(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  __shared__ uint32_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 97u + 13u) & 4095u;
  __syncthreads();
  uint32_t v = seed & 4095u;
  uint64_t t0 = __builtin_readcyclecounter();
#pragma unroll 16
  for (int i = 0; i < 1024; i++) v = s[v];
  uint64_t t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_branch(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t v = seed;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 1024; i++) { v += 1; asm volatile("" : "+v"(v)); if (v & 0x80000000u) { v ^= 5; asm volatile("" : "+v"(v)); } }
  uint64_t t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_global(const uint32_t* chain, uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t v = seed & 65535u;
  uint64_t t0 = __builtin_readcyclecounter();
  for (int i = 0; i < 256; i++) v = chain[v];
  uint64_t t1 = __builtin_readcyclecounter();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* cyc; uint32_t* chain;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64); hipMalloc(&chain, 65536 * 4);
  std::vector<uint32_t> h(65536); for (uint32_t i = 0; i < 65536; i++) h[i] = (i * 40503u + 77u) & 65535u;
  hipMemcpy(chain, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  auto report = [&](const char* name, int ops) { uint64_t c; hipDeviceSynchronize(); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); printf("%-28s %8.1f counter-ticks/op\n", name, (double)c / ops); };
  for (int threads : {1, 64}) {
    printf("-- %d active lane(s)\n", threads);
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_valu, 1, threads, 0, 0, out, cyc, 1u); if (rep) report("dependent v_mad chain", 1024);
      hipLaunchKernelGGL(k_add, 1, threads, 0, 0, out, cyc, 1u); if (rep) report("xor+add pair", 1024);
      hipLaunchKernelGGL(k_shift64, 1, threads, 0, 0, out, cyc, 1u); if (rep) report("64-bit shift chain (3 ops)", 1024);
      hipLaunchKernelGGL(k_lds, 1, threads, 0, 0, out, cyc, 1u); if (rep) report("dependent LDS read", 1024);
      hipLaunchKernelGGL(k_branch, 1, threads, 0, 0, out, cyc, 1u); if (rep) report("loop iter w/ not-taken if", 1024);
      hipLaunchKernelGGL(k_global, 1, threads, 0, 0, chain, out, cyc, 1u); if (rep) report("dependent global (L2) load", 256);
    }
  }
  // wall-clock calibration of the counter: 1 ms busy kernel
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipEventRecord(a); for (int i = 0; i < 200; i++) hipLaunchKernelGGL(k_valu, 1, 64, 0, 0, out, cyc, 1u); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); uint64_t c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  printf("200 launches of the v_mad kernel: %.3f ms wall; %llu ticks per kernel => >= %.1f MHz tick rate if back-to-back\n", ms, (unsigned long long)c, (double)c * 200 / (ms * 1e3));
  return 0;
}
