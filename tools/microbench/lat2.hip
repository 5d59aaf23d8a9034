// Second round: per-instruction cost of dependent VALU ops and of taken branches for one wavefront (gfx950).
#include <hip/hip_runtime.h>
#include <cstdio>
#define KEEP(v) asm volatile("" : "+v"(v))
template <int UNROLL> __global__ void k_chain(uint32_t* out, uint64_t* cyc, uint32_t seed, int iters) {
  uint32_t v = seed + threadIdx.x, w = seed * 7u;
  uint64_t t0 = clock64();
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int j = 0; j < UNROLL; j++) { v = (v >> 1) + w; KEEP(v); }
  }
  uint64_t t1 = clock64();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_sel(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t v = seed + threadIdx.x, w = seed * 7u;
  uint64_t t0 = clock64();
#pragma unroll
  for (int j = 0; j < 512; j++) { v = (v & 1) ? v + w : v ^ w; KEEP(v); }
  uint64_t t1 = clock64();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_mul24(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint32_t v = seed + threadIdx.x;
  uint64_t t0 = clock64();
#pragma unroll
  for (int j = 0; j < 512; j++) { v = (v & 0xFFFu) * 4093u + 1u; KEEP(v); }
  uint64_t t1 = clock64();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_shr64(uint32_t* out, uint64_t* cyc, uint32_t seed) {
  uint64_t v = ((uint64_t)seed << 40) | threadIdx.x | (1ull << 63); uint32_t s = seed & 1;
  uint64_t t0 = clock64();
#pragma unroll
  for (int j = 0; j < 512; j++) { v = (v >> s) | (1ull << 63); asm volatile("" : "+v"(v)); }
  uint64_t t1 = clock64();
  out[threadIdx.x] = (uint32_t)(v >> 20); if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_exec(uint32_t* out, uint64_t* cyc, uint32_t seed) {   // divergent-style if that is always skipped (s_cbranch_execz taken)
  uint32_t v = seed + threadIdx.x, w = seed;
  uint64_t t0 = clock64();
#pragma unroll
  for (int j = 0; j < 256; j++) { v += 1; KEEP(v); if (v == 0xFFFFFFF0u) { v = v * w + 3; v ^= v >> 3; v = v * w + 7; KEEP(v); } }
  uint64_t t1 = clock64();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
__global__ void k_ldsalu(uint32_t* out, uint64_t* cyc, uint32_t seed) {   // LDS read -> 4 ALU -> LDS read ...
  __shared__ uint32_t s[4096];
  for (int i = threadIdx.x; i < 4096; i += blockDim.x) s[i] = (i * 97u + 13u) & 4095u;
  __syncthreads();
  uint32_t v = seed & 4095u;
  uint64_t t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < 512; i++) { v = s[v]; v = (v + 5) & 4095; KEEP(v); v = (v ^ 9) & 4095; KEEP(v); v = (v + 3) & 4095; KEEP(v); v = (v ^ 1) & 4095; KEEP(v); }
  uint64_t t1 = clock64();
  out[threadIdx.x] = v; if (threadIdx.x == 0) cyc[0] = t1 - t0;
}
int main() {
  uint32_t* out; uint64_t* cyc;
  hipMalloc(&out, 4096); hipMalloc(&cyc, 64);
  auto get = [&]() { uint64_t c; hipDeviceSynchronize(); hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost); return (double)c; };
  for (int threads : {1, 64}) {
    printf("-- %d active lane(s)\n", threads);
    for (int rep = 0; rep < 2; rep++) {
      hipLaunchKernelGGL(k_chain<1>, 1, threads, 0, 0, out, cyc, 3u, 1024); double a = get();
      hipLaunchKernelGGL(k_chain<8>, 1, threads, 0, 0, out, cyc, 3u, 1024); double b = get();
      hipLaunchKernelGGL(k_chain<64>, 1, threads, 0, 0, out, cyc, 3u, 1024); double c = get();
      if (rep) printf("loop of 2-op chain: unroll1 %.1f, unroll8 %.1f, unroll64 %.1f ticks/iter => per pair %.2f, per loop-back %.1f\n", a / 1024, b / 1024, c / 1024, (c - b) / 1024 / 56, (8 * a - b) / 1024 / 7);
      hipLaunchKernelGGL(k_sel, 1, threads, 0, 0, out, cyc, 3u); if (rep) printf("and+cmp+add+xor+cndmask group: %.2f ticks\n", get() / 512);
      hipLaunchKernelGGL(k_mul24, 1, threads, 0, 0, out, cyc, 3u); if (rep) printf("and+mul_u24+add group: %.2f ticks\n", get() / 512);
      hipLaunchKernelGGL(k_shr64, 1, threads, 0, 0, out, cyc, 3u); if (rep) printf("lshr64+or64 group: %.2f ticks\n", get() / 512);
      hipLaunchKernelGGL(k_exec, 1, threads, 0, 0, out, cyc, 3u); if (rep) printf("add + skipped if-block: %.2f ticks\n", get() / 256);
      hipLaunchKernelGGL(k_ldsalu, 1, threads, 0, 0, out, cyc, 3u); if (rep) printf("LDS read + 8 ALU: %.2f ticks\n", get() / 512);
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0); hipLaunchKernelGGL(k_chain<64>, 1, 64, 0, 0, out, cyc, 3u, 1 << 20); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); printf("calibration: %.0f ticks in %.3f ms => %.1f MHz\n", get(), ms, get() / ms / 1e3);
  return 0;
}
