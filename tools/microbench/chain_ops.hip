// chain_ops — ns per step of DEPENDENT instruction chains as one lone wavefront of gfx950 sees them (round 6: what the serial entropy chains are made of).
// Every kernel runs one wave64 workgroup through `iters` x 16 unrolled steps of one chain shape; time from HIP events.
//   hipcc --offload-arch=gfx950 -O3 -o chain_ops chain_ops.hip && ./chain_ops
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define STEP16(body) _Pragma("unroll") for (int u = 0; u < 16; u++) { body }

__global__ __launch_bounds__(64) void KVAdd(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x, b = 3;
  for (int it = 0; it < iters; it++) STEP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
  out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void KSAdd(uint32_t* out, int iters) {
  uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x), b = (uint32_t)__builtin_amdgcn_readfirstlane(3);
  for (int it = 0; it < iters; it++) STEP16(asm volatile("s_add_u32 %0, %0, %1" : "+s"(a) : "s"(b) : "scc");)
  out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void KVMulLo(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x | 1, b = 3;
  for (int it = 0; it < iters; it++) STEP16(asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a) : "v"(b));)
  out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void KVMad24(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x | 1, b = 3;
  for (int it = 0; it < iters; it++) STEP16(asm volatile("v_mad_u32_u24 %0, %0, %1, %1" : "+v"(a) : "v"(b));)
  out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void KSMul(uint32_t* out, int iters) {
  uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x | 1), b = (uint32_t)__builtin_amdgcn_readfirstlane(3);
  for (int it = 0; it < iters; it++) STEP16(asm volatile("s_mul_i32 %0, %0, %1" : "+s"(a) : "s"(b));)
  out[threadIdx.x] = a;
}
// LDS pointer chase: lds[i] holds the byte offset of the next slot
__global__ __launch_bounds__(64) void KLds(uint32_t* out, int iters) {
  __shared__ uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 37 + 11) & 1023) * 4;
  __syncthreads();
  uint32_t a = threadIdx.x * 4;
  for (int it = 0; it < iters; it++) STEP16(a = *(volatile uint32_t*)((char*)lds + a);)
  out[threadIdx.x] = a;
}
// LDS round trip + 2 VALU ops on the chain (address arithmetic)
__global__ __launch_bounds__(64) void KLdsPlus2(uint32_t* out, int iters) {
  __shared__ uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (i * 37 + 11) & 1023;
  __syncthreads();
  uint32_t a = threadIdx.x;
  for (int it = 0; it < iters; it++) STEP16(a = *(volatile uint32_t*)((char*)lds + ((a & 1023) << 2));)
  out[threadIdx.x] = a;
}
// v_readlane chain: lane index comes from the previous read
__global__ __launch_bounds__(64) void KReadlane(uint32_t* out, int iters) {
  const uint32_t v = (threadIdx.x * 37 + 11) & 63;
  uint32_t s = 0;
  for (int it = 0; it < iters; it++) STEP16(s = (uint32_t)__builtin_amdgcn_readlane((int)v, (int)s);)
  out[threadIdx.x] = s;
}
// ballot -> popcount -> readlane -> (back to a VGPR compare): the "tree as thresholds in lanes" step
__global__ __launch_bounds__(64) void KBallotWalk(uint32_t* out, int iters) {
  const int32_t thr = (int32_t)threadIdx.x * 8 - 256;
  const int32_t leafv = (int32_t)((threadIdx.x * 37 + 11) & 511) - 256;
  int32_t v = 5;
  for (int it = 0; it < iters; it++) STEP16(
    const uint64_t m = __ballot(v > thr);
    const int k = __builtin_popcountll(m) & 63;
    v = __builtin_amdgcn_readlane(leafv, k);
  )
  out[threadIdx.x] = (uint32_t)v;
}
// readfirstlane round trip: VGPR -> SGPR -> VGPR op
__global__ __launch_bounds__(64) void KReadfirst(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x;
  for (int it = 0; it < iters; it++) STEP16(
    uint32_t s = (uint32_t)__builtin_amdgcn_readfirstlane((int)a);
    asm volatile("v_add_u32 %0, %1, %2" : "=v"(a) : "s"(s), "v"(1u));
  )
  out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void KBpermute(uint32_t* out, int iters) {
  uint32_t a = ((threadIdx.x * 37 + 11) & 63) * 4;
  for (int it = 0; it < iters; it++) STEP16(a = (uint32_t)__builtin_amdgcn_ds_bpermute((int)a, (int)a);)
  out[threadIdx.x] = a;
}
__global__ __launch_bounds__(64) void KGlobal(uint32_t* out, const uint32_t* __restrict__ tab, int iters) {
  uint32_t a = threadIdx.x;
  for (int it = 0; it < iters; it++) STEP16(a = __builtin_nontemporal_load(tab + (a & 1023));)
  out[threadIdx.x] = a;
}
// scalar load chase (constant cache)
__global__ __launch_bounds__(64) void KSLoad(uint32_t* out, const uint32_t* __restrict__ tab, int iters) {
  uint32_t a = (uint32_t)__builtin_amdgcn_readfirstlane((int)blockIdx.x);
  for (int it = 0; it < iters; it++) STEP16(a = (uint32_t)__builtin_amdgcn_readfirstlane((int)tab[a & 1023]);)
  out[threadIdx.x] = a;
}
// 64-bit shift on the vector unit and on the scalar unit
__global__ __launch_bounds__(64) void KVShr64(uint32_t* out, int iters) {
  uint64_t a = 0x123456789abcdefull + threadIdx.x; uint32_t n = 1;
  for (int it = 0; it < iters; it++) STEP16(asm volatile("v_lshrrev_b64 %0, %1, %0" : "+v"(a) : "v"(n)); a |= 0x8000000000000000ull;)
  out[threadIdx.x] = (uint32_t)a;
}
// mixed: two independent dependent chains interleaved (does a lone wave overlap them?)
__global__ __launch_bounds__(64) void KVAdd2(uint32_t* out, int iters) {
  uint32_t a = threadIdx.x, c = threadIdx.x + 7, b = 3;
  for (int it = 0; it < iters; it++) STEP16(asm volatile("v_add_u32 %0, %0, %1" : "+v"(a) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c) : "v"(b));)
  out[threadIdx.x] = a + c;
}
// LDS chase with independent VALU work in its shadow: 8 independent v_adds per LDS round trip
__global__ __launch_bounds__(64) void KLdsShadow(uint32_t* out, int iters) {
  __shared__ uint32_t lds[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = ((i * 37 + 11) & 1023) * 4;
  __syncthreads();
  uint32_t a = threadIdx.x * 4, c0 = 1, c1 = 2, c2 = 3, c3 = 4, b = 3;
  for (int it = 0; it < iters; it++) STEP16(
    a = *(volatile uint32_t*)((char*)lds + a);
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(c0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c1) : "v"(b));
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(c2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c3) : "v"(b));
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(c0) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c1) : "v"(b));
    asm volatile("v_add_u32 %0, %0, %1" : "+v"(c2) : "v"(b)); asm volatile("v_add_u32 %0, %0, %1" : "+v"(c3) : "v"(b));
  )
  out[threadIdx.x] = a + c0 + c1 + c2 + c3;
}

template <typename F> static void Time(const char* name, int steps_per_iter, int iters, F launch) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch(iters / 8);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  launch(iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  printf("%-40s %8.2f ns per step\n", name, ms * 1e6 / ((double)iters * steps_per_iter));
}

int main() {
  uint32_t *out, *tab;
  hipMalloc(&out, 4096); hipMalloc(&tab, 4096);
  uint32_t h[1024];
  for (int i = 0; i < 1024; i++) h[i] = (i * 37 + 11) & 1023;
  hipMemcpy(tab, h, 4096, hipMemcpyHostToDevice);
  const int N = 200000;
  Time("v_add_u32 dependent", 16, N, [&](int n) { KVAdd<<<1, 64>>>(out, n); });
  Time("2 x v_add_u32 chains interleaved (pair)", 16, N, [&](int n) { KVAdd2<<<1, 64>>>(out, n); });
  Time("s_add_u32 dependent", 16, N, [&](int n) { KSAdd<<<1, 64>>>(out, n); });
  Time("v_mul_lo_u32 dependent", 16, N, [&](int n) { KVMulLo<<<1, 64>>>(out, n); });
  Time("v_mad_u32_u24 dependent", 16, N, [&](int n) { KVMad24<<<1, 64>>>(out, n); });
  Time("s_mul_i32 dependent", 16, N, [&](int n) { KSMul<<<1, 64>>>(out, n); });
  Time("v_lshrrev_b64 (+or) dependent", 16, N, [&](int n) { KVShr64<<<1, 64>>>(out, n); });
  Time("LDS pointer chase (ds_read_b32)", 16, N, [&](int n) { KLds<<<1, 64>>>(out, n); });
  Time("LDS chase + and + shl", 16, N, [&](int n) { KLdsPlus2<<<1, 64>>>(out, n); });
  Time("LDS chase with 8 v_add in its shadow", 16, N, [&](int n) { KLdsShadow<<<1, 64>>>(out, n); });
  Time("v_readlane chain (SGPR index)", 16, N, [&](int n) { KReadlane<<<1, 64>>>(out, n); });
  Time("cmp/ballot + bcnt + readlane", 16, N, [&](int n) { KBallotWalk<<<1, 64>>>(out, n); });
  Time("readfirstlane + v_add", 16, N, [&](int n) { KReadfirst<<<1, 64>>>(out, n); });
  Time("ds_bpermute chain", 16, N, [&](int n) { KBpermute<<<1, 64>>>(out, n); });
  Time("global load chase (L1/L2 hit)", 16, N / 4, [&](int n) { KGlobal<<<1, 64>>>(out, tab, n); });
  Time("scalar load chase (K$)", 16, N / 4, [&](int n) { KSLoad<<<1, 64>>>(out, tab, n); });
  return 0;
}
