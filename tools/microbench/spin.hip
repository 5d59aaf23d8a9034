// Co-runner for interference experiments (tools/experiments/gpu_corun.py): long-lived wavefronts that do a configurable kind of work for a given time,
// launched beside a real decode stage to see which shared resource that stage is short of.
//   mode 0: dependent VALU chain only            mode 1: + `lds` bytes of LDS held per workgroup
//   mode 2: + one scattered 4-byte load per lane and round over `span` bytes (TLB / L2 pressure)      mode 3: scattered 4-byte stores instead
#include <hip/hip_runtime.h>
#include <stdint.h>
extern __shared__ uint8_t dyn[];
__global__ void SpinKernel(uint32_t* buf, uint64_t span_words, int mode, uint64_t ticks, int active_lanes, int prio, uint32_t* sink) {
  if (prio) __builtin_amdgcn_s_setprio(3);
  const uint64_t t0 = wall_clock64();
  uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 1u;
  uint32_t acc = 0;
  const bool on = (int)(threadIdx.x & 63) < active_lanes;
  while (wall_clock64() - t0 < ticks) {
#pragma unroll 1
    for (int i = 0; i < 64; i++) {                       // ~ a lock-step decode iteration's worth of dependent integer work
      x = x * 1664525u + 1013904223u; acc ^= x >> 7; x += acc & 3u;
    }
    if (on && mode == 2) acc += buf[((uint64_t)x * 2654435761ull) % span_words];
    if (on && mode == 3) buf[((uint64_t)x * 2654435761ull) % span_words] = acc;
    if (mode == 1) dyn[threadIdx.x] = (uint8_t)acc;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
extern "C" int spin_launch(void* stream, int blocks, int threads, int lds, void* buf, uint64_t span_bytes, int mode, double ms, int active_lanes, int prio, void* sink) {
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute((const void*)SpinKernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048); attr = true; }
  hipLaunchKernelGGL(SpinKernel, dim3(blocks), dim3(threads), lds, (hipStream_t)stream, (uint32_t*)buf, span_bytes / 4, mode, (uint64_t)(ms * 1e5), active_lanes, prio, (uint32_t*)sink);
  return (int)hipGetLastError();
}
