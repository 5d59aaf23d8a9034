// wave_ans — ns per sample of the wave-wide ANS + gradient-context chain (kernels.hip WaveSegment) on synthetic tables and a random bit stream, one lone
// wavefront; variants of the loop are timed against each other and checked against a host-side decode of the same stream.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o wave_ans wave_ans.hip && ./wave_ans
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <algorithm>

extern __shared__ __align__(16) uint8_t g_dyn_lds[];
template <typename T> __device__ __forceinline__ T LdG(const T* p) { return *p; }
template <typename T> __device__ __forceinline__ void StG(T* p, T v) { *p = v; }
template <typename T> __device__ __forceinline__ T LdS(uint32_t byte_off) { return *reinterpret_cast<const T*>(g_dyn_lds + byte_off); }
template <typename T> __device__ __forceinline__ void StS(uint32_t byte_off, T v) { *reinterpret_cast<T*>(g_dyn_lds + byte_off) = v; }
__device__ __forceinline__ uint32_t Uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__host__ __device__ inline int32_t UnpackSigned(uint32_t u) { return (int32_t)((u >> 1) ^ (~(u & 1) + 1)); }
constexpr int32_t kWideEscape = -128;

#include "wave_ans_core.h"

constexpr uint32_t kLa = 6, kClusters = 18, kSlots = 1u << kLa, kCfg = 4 | (1 << 8) | (0 << 16);   // hybrid uint config: split_exponent 4, msb 1, lsb 0
struct Tables { std::vector<uint64_t> wide; std::vector<uint16_t> cut; };

// alias table of one distribution (sum 4096), dec_ans.h / host_parse.cc BuildAlias
static void BuildAlias(std::vector<int> dist, uint32_t cluster, Tables& t) {
  const int T = kSlots, B = 4096 / T;
  dist.resize(T, 0);
  std::vector<int> cut(T), right(T, 0), offs1(T, 0), over, under;
  for (int i = 0; i < T; i++) cut[i] = dist[i];
  for (int i = 0; i < T; i++) { if (cut[i] > B) over.push_back(i); else if (cut[i] < B) under.push_back(i); }
  while (!over.empty()) {
    int o = over.back(); over.pop_back();
    int u = under.back(); under.pop_back();
    cut[o] -= B - cut[u];
    right[u] = o; offs1[u] = cut[o];
    if (cut[o] < B) under.push_back(o); else if (cut[o] > B) over.push_back(o);
  }
  auto wv = [](uint32_t tok) { const uint32_t split = 1u << (kCfg & 0xFF); return (tok < split && tok < 255u) ? ((uint32_t)UnpackSigned(tok) & 0xFFu) : 0x80u; };
  for (int i = 0; i < T; i++) {
    if (cut[i] == B) { right[i] = i; offs1[i] = 0; cut[i] = 0; } else offs1[i] -= cut[i];
    const uint32_t f0 = std::max(dist[i], 1) - 1, f1 = std::max(dist[right[i]], 1) - 1;
    const uint32_t lo = f0 | (wv(i) << 24), hi = f1 | ((uint32_t)offs1[i] << 12) | (wv(right[i]) << 24);
    t.wide[cluster * kSlots + i] = (uint64_t)lo | ((uint64_t)hi << 32);
    t.cut[cluster * kSlots + i] = (uint16_t)(cut[i] | (right[i] << 8));
  }
}

template <int V> __global__ __launch_bounds__(64) void Decode(const uint32_t* words, uint32_t wend, int32_t* out, int rows, const int32_t* thr, const uint64_t* wide, const uint16_t* cut, uint32_t nthr) {
  const uint32_t lane = threadIdx.x;
  const uint32_t wide_off = 0, cut_off = kClusters * kSlots * 8;
  for (uint32_t i = lane; i < kClusters * kSlots; i += 64) { StS<uint64_t>(wide_off + i * 8, wide[i]); StS<uint16_t>(cut_off + i * 2, cut[i]); }
  __syncthreads();
  WaveBits bits; bits.Start(words, Uniform(wend), 32, lane);
  WaveChan wc;
  wc.thr = lane < nthr ? thr[lane] : 0x7FFFFFFF;
  wc.cluster = min(lane, nthr);
  wc.la = Uniform(kLa); wc.cfg_off = 0; wc.cfg_uniform = Uniform(kCfg);
  wc.abase = wide_off + ((wc.cluster << kLa) << 3); wc.cbase = cut_off + ((wc.cluster << kLa) << 1);
  uint32_t state = Uniform(words[0] | 0x10000u);
  int32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;
  for (int y = 0; y < rows; y++) {
    int32_t left = 0, nw = 0;
    if (y > 0) { left = __builtin_amdgcn_readlane(p0, 0); nw = left; }
    int32_t c[4] = {0, 0, 0, 0};
#pragma unroll
    for (int seg = 0; seg < 4; seg++) {
      const int32_t prevv = seg == 0 ? p0 : seg == 1 ? p1 : seg == 2 ? p2 : p3;
      int32_t curv = 0;
      if (y > 0) WaveSegmentV<V, true, true, 5>(bits, state, left, nw, prevv, curv, 64, wc);
      else WaveSegmentV<V, false, true, 5>(bits, state, left, nw, prevv, curv, 64, wc);
      out[(size_t)y * 256 + seg * 64 + lane] = curv;
      c[seg] = curv;
    }
    p0 = c[0]; p1 = c[1]; p2 = c[2]; p3 = c[3];
  }
  if (lane == 0) { out[(size_t)rows * 256] = (int32_t)state; out[(size_t)rows * 256 + 1] = (int32_t)bits.BitPos(); }
}

// host reference of the same decode
static void HostDecode(const std::vector<uint32_t>& words, int rows, const std::vector<int32_t>& thr, const Tables& t, std::vector<int32_t>& out) {
  uint64_t bitpos = 32;
  auto read = [&](int n) { uint64_t v = 0; for (int i = 0; i < n; i++) { const uint64_t b = (words[(bitpos >> 5)] >> (bitpos & 31)) & 1; v |= b << i; bitpos++; } return (uint32_t)v; };
  uint32_t state = words[0] | 0x10000u;
  out.assign((size_t)rows * 256 + 2, 0);
  for (int y = 0; y < rows; y++) for (int x = 0; x < 256; x++) {
    const int32_t* row = &out[(size_t)y * 256];
    int32_t W = x ? row[x - 1] : (y ? row[x - 256] : 0), N = y ? row[x - 256] : W, NW = (x && y) ? row[x - 257] : W;
    const int32_t v0 = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
    uint32_t k = 0; for (int32_t c : thr) k += v0 > c;
    const int32_t m = std::min(N, W), M = std::max(N, W), guess = std::max(m, std::min(M, v0));
    const uint32_t slot = (state & 0xFFF) >> (12 - kLa), pos = state & ((1u << (12 - kLa)) - 1), hi = state >> 12;
    const uint64_t e = t.wide[k * kSlots + slot]; const uint32_t cr = t.cut[k * kSlots + slot];
    const bool hit = pos >= (cr & 0xFF);
    const uint32_t sw = hit ? (uint32_t)(e >> 32) : (uint32_t)e;
    state = (sw & 0xFFF) * hi + hi + pos + ((sw >> 12) & 0xFFF);
    uint32_t tok = hit ? (cr >> 8) : slot;
    if (state < (1u << 16)) state = (state << 16) | read(16);
    const uint32_t split_exp = kCfg & 0xFF, split = 1u << split_exp;
    if (tok >= split) {
      const uint32_t msb = (kCfg >> 8) & 0xFF, lsb = (kCfg >> 16) & 0xFF;
      const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
      const uint32_t low = tok & ((1u << lsb) - 1);
      tok >>= lsb;
      const uint32_t xb = read((int)nbits);
      const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
      tok = (((hb << nbits) | xb) << lsb) | low;
    }
    out[(size_t)y * 256 + x] = (int32_t)((uint32_t)UnpackSigned(tok) + (uint32_t)guess);
  }
  out[(size_t)rows * 256] = (int32_t)state; out[(size_t)rows * 256 + 1] = (int32_t)bitpos;
}

template <int V> static void Run(const char* name, const uint32_t* dwords, uint32_t nwords, int32_t* dout, int rows, const int32_t* dthr, const uint64_t* dwide, const uint16_t* dcut, uint32_t nthr, const std::vector<int32_t>& ref) {
  const uint32_t lds = kClusters * kSlots * 10;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  Decode<V><<<1, 64, lds>>>(dwords, nwords, dout, rows, dthr, dwide, dcut, nthr);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  Decode<V><<<1, 64, lds>>>(dwords, nwords, dout, rows, dthr, dwide, dcut, nthr);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
  std::vector<int32_t> got(ref.size());
  (void)hipMemcpy(got.data(), dout, got.size() * 4, hipMemcpyDeviceToHost);
  size_t bad = 0; for (size_t i = 0; i < ref.size(); i++) bad += got[i] != ref[i];
  printf("%-28s %7.1f ns per sample   (%s: %zu of %zu values differ; end state %08x bit %d)\n", name, ms * 1e6 / ((double)rows * 256), bad ? "MISMATCH" : "ok", bad, ref.size(), (uint32_t)got[ref.size() - 2], got[ref.size() - 1]);
}

int main(int argc, char** argv) {
  const int rows = argc > 1 ? atoi(argv[1]) : 512;
  const double decay = argc > 2 ? atof(argv[2]) : 0.55;        // geometric token distribution: P(tok = i) ~ decay^i
  Tables t; t.wide.resize(kClusters * kSlots); t.cut.resize(kClusters * kSlots);
  for (uint32_t c = 0; c < kClusters; c++) {
    std::vector<int> dist(40, 0);
    double r = decay + 0.02 * (c % 5), p = 1.0, sum = 0;
    std::vector<double> w(40); for (int i = 0; i < 40; i++) { w[i] = p; sum += p; p *= r; }
    int total = 0; for (int i = 0; i < 40; i++) { dist[i] = std::max(1, (int)(w[i] / sum * 4096)); total += dist[i]; }
    dist[0] += 4096 - total;
    BuildAlias(dist, c, t);
  }
  std::vector<int32_t> thr = {-255, -127, -63, -31, -15, -7, -3, -1, 0, 1, 3, 7, 15, 31, 63, 127, 255};
  const uint32_t nwords = (uint32_t)rows * 256 * 2 + 256;
  std::vector<uint32_t> words(nwords);
  uint64_t s = 0x9E3779B97F4A7C15ull; for (auto& w : words) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; w = (uint32_t)(s >> 16); }
  std::vector<int32_t> ref; HostDecode(words, rows, thr, t, ref);
  size_t esc = 0; { /* escape rate of the reference run is not tracked per sample: estimate from the values */ for (size_t i = 0; i + 2 < ref.size(); i++) esc += 0; }
  uint32_t* dwords; int32_t *dout, *dthr; uint64_t* dwide; uint16_t* dcut;
  (void)hipMalloc(&dwords, nwords * 4); (void)hipMalloc(&dout, ref.size() * 4); (void)hipMalloc(&dthr, 64 * 4); (void)hipMalloc(&dwide, t.wide.size() * 8); (void)hipMalloc(&dcut, t.cut.size() * 2);
  (void)hipMemcpy(dwords, words.data(), nwords * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dthr, thr.data(), thr.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(dwide, t.wide.data(), t.wide.size() * 8, hipMemcpyHostToDevice);
  (void)hipMemcpy(dcut, t.cut.data(), t.cut.size() * 2, hipMemcpyHostToDevice);
  printf("rows %d (x 256 samples), %u clusters, decay %.2f, bits used %d (%.2f per sample)\n", rows, kClusters, decay, ref[ref.size() - 1], ref[ref.size() - 1] / ((double)rows * 256));
  Run<0>("compiled C++ (kernels.hip)", dwords, nwords, dout, rows, dthr, dwide, dcut, (uint32_t)thr.size(), ref);
  Run<1>("variant 1", dwords, nwords, dout, rows, dthr, dwide, dcut, (uint32_t)thr.size(), ref);
  Run<2>("variant 2", dwords, nwords, dout, rows, dthr, dwide, dcut, (uint32_t)thr.size(), ref);
  return 0;
}
