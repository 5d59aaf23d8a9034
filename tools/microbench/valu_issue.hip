// valu_issue — cycles per wave64 VALU instruction on gfx950 (the bench line's `valu_issue` peak assumed 4 cycles, the CDNA4 guide says 2).
//
// Every wave runs a long unrolled loop of independent v_fma_f32 (or v_pk_fma_f32) chains — ILP 1, 2, 4, 8 accumulators — and reads the shader clock
// (s_memtime) before and after; W = 1, 2, 4, 8 waves per SIMD (W workgroups of 256 threads per CU).  Reported: cycles per instruction as ONE wave sees it
// and per SIMD (all its waves together), and from the wall clock of the launch the chip-wide rate in wave-instructions per second.
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>

typedef float float2v __attribute__((ext_vector_type(2)));

template <int ILP, bool PACKED>
__global__ __launch_bounds__(256) void Fma(float* __restrict__ out, uint64_t* __restrict__ cycles, int iters) {
  float a[8]; float2v p[8];
  const float b = 1.0000001f, c = 1e-9f;
  const float2v b2 = {b, b}, c2 = {c, c};
  for (int i = 0; i < 8; i++) { a[i] = (float)(threadIdx.x + i); p[i] = float2v{a[i], a[i] + 1.0f}; }
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 64 / ILP; u++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) {
        if (PACKED) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(b2), "v"(c2));
        else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(b), "v"(c));
      }
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 8; i++) s += a[i] + p[i].x + p[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

// the same for the scalar unit: dependent / independent s_add_u32 chains of ONE wave per SIMD (what a serial entropy-decode chain moved to SALU would issue at)
template <int ILP>
__global__ __launch_bounds__(256) void Salu(float* __restrict__ out, uint64_t* __restrict__ cycles, int iters) {
  uint32_t a[8];
  uint32_t b = 3;
  for (int i = 0; i < 8; i++) a[i] = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x + i));
  b = (uint32_t)__builtin_amdgcn_readfirstlane((int)b);
  const uint64_t t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; it++) {
#pragma unroll
    for (int u = 0; u < 64 / ILP; u++) {
#pragma unroll
      for (int i = 0; i < ILP; i++) asm volatile("s_add_u32 %0, %0, %1" : "+s"(a[i]) : "s"(b) : "scc");
    }
  }
  const uint64_t t1 = __builtin_readcyclecounter();
  uint32_t s = 0;
  for (int i = 0; i < 8; i++) s += a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = (float)s;
  if ((threadIdx.x & 63) == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}
template <int ILP> void RunSalu(int waves_per_simd, int num_cu, float* out, uint64_t* cyc) {
  const int blocks = num_cu * waves_per_simd, iters = 2000;
  hipLaunchKernelGGL((Salu<ILP>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  std::vector<uint64_t> h((size_t)blocks * 4);
  (void)hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  printf("{\"salu\": 1, \"ilp\": %d, \"waves_per_simd\": %d, \"cycles_per_instr_one_wave\": %.2f}\n", ILP, waves_per_simd, (double)h[h.size() / 2] / (64.0 * iters));
}

template <int ILP, bool PACKED> void Run(int waves_per_simd, int num_cu, float* out, uint64_t* cyc) {
  const int blocks = num_cu * waves_per_simd, iters = 2000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((Fma<ILP, PACKED>), dim3(blocks), dim3(256), 0, 0, out, cyc, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((Fma<ILP, PACKED>), dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<uint64_t> h((size_t)blocks * 4);
  hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
  std::sort(h.begin(), h.end());
  const double instr = 64.0 * iters;                                   // wave-instructions per wave
  const double med = (double)h[h.size() / 2];
  const double total = instr * blocks * 4;
  printf("{\"packed\": %d, \"ilp\": %d, \"waves_per_simd\": %d, \"cycles_per_instr_one_wave\": %.2f, \"cycles_per_instr_per_simd\": %.2f, \"launch_ms\": %.3f, \"wave_instr_per_s\": %.4g, "
         "\"counter_cycles_per_wave\": %.0f}\n", PACKED ? 1 : 0, ILP, waves_per_simd, med / instr, med / instr / waves_per_simd, ms, total / (ms * 1e-3), med);
}

int main() {
  hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
  const int num_cu = prop.multiProcessorCount;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_khz\": %d, \"note\": \"__builtin_readcyclecounter = s_memtime: ticks of the shader clock per the CDNA4 guide; wave_instr_per_s comes from HIP events\"}\n", prop.name, num_cu, prop.clockRate);
  float* out; uint64_t* cyc;
  hipMalloc(&out, (size_t)num_cu * 8 * 256 * 4); hipMalloc(&cyc, (size_t)num_cu * 8 * 4 * 8);
  for (int w : {1, 2, 4, 8}) {
    Run<1, false>(w, num_cu, out, cyc); Run<2, false>(w, num_cu, out, cyc); Run<4, false>(w, num_cu, out, cyc); Run<8, false>(w, num_cu, out, cyc);
    Run<1, true>(w, num_cu, out, cyc); Run<4, true>(w, num_cu, out, cyc); Run<8, true>(w, num_cu, out, cyc);
    if (w <= 2) { RunSalu<1>(w, num_cu, out, cyc); RunSalu<8>(w, num_cu, out, cyc); }
  }
  return 0;
}
