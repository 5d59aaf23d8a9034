// jxl-hip: hand-written HIP kernels for gfx950 (MI355X) — the JPEG XL decode hot path.
// Stages: K_lf (LF coefficients + HF metadata entropy decode, b3/b4), K_lfpost (LF dequant, adaptive
// smoothing, LLF, b6), K_hf (ANS coefficient decode, b3/b7), K_idct (dequant + CfL + variable-block IDCT, b8/b9),
// K_gab (b11), K_epf (b12), K_out (XYB -> linear -> sRGB -> interleaved write, b15-b17), K_mod* (Modular, b4/b5).
// Compiled with -ffp-contract=off: every fused multiply-add is an explicit fmaf so results are bit-identical to the
// CPU oracle's (and to what libjxl's MulAdd does on FMA hardware).
#include "kernels.h"
#include <hip/hip_runtime.h>
#include <math.h>
#include <algorithm>
#include <mutex>

namespace jxlhip {

__constant__ float d_wc[9][128];       // WcMultipliers<N>[i] = 1 / (2 cos((i + 0.5) pi / N)), row = log2 N
__constant__ float d_resample[6][32];  // DCTTotalResampleScale<N, 8N>(k), row = log2 N
// dec_transforms-inl.h k4x4AFVBasis: the 16 orthonormal basis functions of the AFV 4x4 corner transform
__constant__ float d_afv_basis[16][16] = {
    {0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f, 0.25f},
    {0.876902929799142f, 0.2206518106944235f, -0.10140050393753763f, -0.1014005039375375f, 0.2206518106944236f, -0.10140050393753777f, -0.10140050393753772f, -0.10140050393753763f, -0.10140050393753758f, -0.10140050393753769f, -0.1014005039375375f, -0.10140050393753768f, -0.10140050393753768f, -0.10140050393753759f, -0.10140050393753763f, -0.10140050393753741f},
    {0.0f, 0.0f, 0.40670075830260755f, 0.44444816619734445f, 0.0f, 0.0f, 0.19574399372042936f, 0.2929100136981264f, -0.40670075830260716f, -0.19574399372042872f, 0.0f, 0.11379074460448091f, -0.44444816619734384f, -0.29291001369812636f, -0.1137907446044814f, 0.0f},
    {0.0f, 0.0f, -0.21255748058288748f, 0.3085497062849767f, 0.0f, 0.4706702258572536f, -0.1621205195722993f, 0.0f, -0.21255748058287047f, -0.16212051957228327f, -0.47067022585725277f, -0.1464291867126764f, 0.3085497062849487f, 0.0f, -0.14642918671266536f, 0.4251149611657548f},
    {0.0f, -0.7071067811865474f, 0.0f, 0.0f, 0.7071067811865476f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f},
    {-0.4105377591765233f, 0.6235485373547691f, -0.06435071657946274f, -0.06435071657946266f, 0.6235485373547694f, -0.06435071657946284f, -0.0643507165794628f, -0.06435071657946274f, -0.06435071657946272f, -0.06435071657946279f, -0.06435071657946266f, -0.06435071657946277f, -0.06435071657946277f, -0.06435071657946273f, -0.06435071657946274f, -0.0643507165794626f},
    {0.0f, 0.0f, -0.4517556589999482f, 0.15854503551840063f, 0.0f, -0.04038515160822202f, 0.0074182263792423875f, 0.39351034269210167f, -0.45175565899994635f, 0.007418226379244351f, 0.1107416575309343f, 0.08298163094882051f, 0.15854503551839705f, 0.3935103426921022f, 0.0829816309488214f, -0.45175565899994796f},
    {0.0f, 0.0f, -0.304684750724869f, 0.5112616136591823f, 0.0f, 0.0f, -0.290480129728998f, -0.06578701549142804f, 0.304684750724884f, 0.2904801297290076f, 0.0f, -0.23889773523344604f, -0.5112616136592012f, 0.06578701549142545f, 0.23889773523345467f, 0.0f},
    {0.0f, 0.0f, 0.3017929516615495f, 0.25792362796341184f, 0.0f, 0.16272340142866204f, 0.09520022653475037f, 0.0f, 0.3017929516615503f, 0.09520022653475055f, -0.16272340142866173f, -0.35312385449816297f, 0.25792362796341295f, 0.0f, -0.3531238544981624f, -0.6035859033230976f},
    {0.0f, 0.0f, 0.40824829046386274f, 0.0f, 0.0f, 0.0f, 0.0f, -0.4082482904638628f, -0.4082482904638635f, 0.0f, 0.0f, -0.40824829046386296f, 0.0f, 0.4082482904638634f, 0.408248290463863f, 0.0f},
    {0.0f, 0.0f, 0.1747866975480809f, 0.0812611176717539f, 0.0f, 0.0f, -0.3675398009862027f, -0.307882213957909f, -0.17478669754808135f, 0.3675398009862011f, 0.0f, 0.4826689115059883f, -0.08126111767175039f, 0.30788221395790305f, -0.48266891150598584f, 0.0f},
    {0.0f, 0.0f, -0.21105601049335784f, 0.18567180916109802f, 0.0f, 0.0f, 0.49215859013738733f, -0.38525013709251915f, 0.21105601049335806f, -0.49215859013738905f, 0.0f, 0.17419412659916217f, -0.18567180916109904f, 0.3852501370925211f, -0.1741941265991621f, 0.0f},
    {0.0f, 0.0f, -0.14266084808807264f, -0.3416446842253372f, 0.0f, 0.7367497537172237f, 0.24627107722075148f, -0.08574019035519306f, -0.14266084808807344f, 0.24627107722075137f, 0.14883399227113567f, -0.04768680350229251f, -0.3416446842253373f, -0.08574019035519267f, -0.047686803502292804f, -0.14266084808807242f},
    {0.0f, 0.0f, -0.13813540350758585f, 0.3302282550303788f, 0.0f, 0.08755115000587084f, -0.07946706605909573f, -0.4613374887461511f, -0.13813540350758294f, -0.07946706605910261f, 0.49724647109535086f, 0.12538059448563663f, 0.3302282550303805f, -0.4613374887461554f, 0.12538059448564315f, -0.13813540350758452f},
    {0.0f, 0.0f, -0.17437602599651067f, 0.0702790691196284f, 0.0f, -0.2921026642334881f, 0.3623817333531167f, 0.0f, -0.1743760259965108f, 0.36238173335311646f, 0.29210266423348785f, -0.4326608024727445f, 0.07027906911962818f, 0.0f, -0.4326608024727457f, 0.34875205199302267f},
    {0.0f, 0.0f, 0.11354987314994337f, -0.07417504595810355f, 0.0f, 0.19402893032594343f, -0.435190496523228f, 0.21918684838857466f, 0.11354987314994257f, -0.4351904965232251f, 0.5550443808910661f, -0.25468277124066463f, -0.07417504595810233f, 0.2191868483885728f, -0.25468277124066413f, 0.1135498731499429f},
};

__device__ __forceinline__ void SetError(const FrameDev& f, uint32_t e) { atomicOr(f.status, e); }
// A frame whose LF stage failed has no usable block info, varblock lists or coefficient offsets — whatever the arena held before is still there (stale values of another
// layout after a refill, anything after a pooled allocation) — so every stage behind the LF decode that indexes through them leaves such a frame alone (round 4: a damaged
// LF stream + a poisoned arena made LlfKernel read a varblock count of 0xA5A5A5A5: memory fault).
__device__ __forceinline__ bool FrameFailed(const FrameDev& f) { return __hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0; }

// =====================================================================================================================
// small device-side field readers (GroupHeader / transforms of modular sub-streams)
// =====================================================================================================================
struct U32D { int bits; uint32_t off; };
__device__ __forceinline__ uint32_t ReadU32(BitReader& br, U32D d0, U32D d1, U32D d2, U32D d3) {
  const uint32_t s = br.Read(2);
  const U32D d = s == 0 ? d0 : s == 1 ? d1 : s == 2 ? d2 : d3;
  return d.off + (d.bits ? br.Read(d.bits) : 0);
}
__device__ __forceinline__ int CeilLog2D(uint32_t x) { return x <= 1 ? 0 : 32 - __clz((int)(x - 1)); }

struct GroupHeaderD {
  uint32_t use_global_tree;
  WPHeader wp;
  uint32_t ntransforms;
  struct { uint32_t id, begin_c, rct_type, num_c, nb_colors, nb_deltas, predictor; } t[8];
};

__device__ bool ReadGroupHeader(BitReader& br, GroupHeaderD& gh) {
  gh.use_global_tree = br.Read(1);
  gh.wp = WPHeader{16, 10, {7, 7, 7, 0, 0}, {13, 12, 12, 12}};
  if (!br.Read(1)) {
    gh.wp.p1 = br.Read(5); gh.wp.p2 = br.Read(5);
    for (int i = 0; i < 5; i++) gh.wp.p3[i] = br.Read(5);
    for (int i = 0; i < 4; i++) gh.wp.w[i] = br.Read(4);
  }
  gh.ntransforms = ReadU32(br, {0, 0}, {0, 1}, {4, 2}, {8, 18});
  if (gh.ntransforms > 8) return false;
  for (uint32_t i = 0; i < gh.ntransforms; i++) {
    auto& t = gh.t[i];
    t.id = br.Read(2);
    if (t.id >= 2) return false;  // squeeze / invalid: unsupported on device
    t.begin_c = ReadU32(br, {3, 0}, {6, 8}, {10, 72}, {13, 1096});
    if (t.id == 0) { t.rct_type = ReadU32(br, {0, 6}, {2, 0}, {4, 2}, {6, 10}); if (t.rct_type >= 42) return false; }
    else {
      t.num_c = ReadU32(br, {0, 1}, {0, 3}, {0, 4}, {13, 1});
      t.nb_colors = ReadU32(br, {8, 0}, {10, 256}, {12, 1280}, {16, 5376});
      t.nb_deltas = ReadU32(br, {0, 0}, {8, 1}, {10, 257}, {16, 1281});
      t.predictor = br.Read(4);
      if (t.nb_deltas != 0 || t.predictor != 0) return false;  // delta palettes: unsupported on device
    }
  }
  return true;
}

// =====================================================================================================================
// Fast serial entropy decode: LDS-resident tables + prefetching bit reader.
// Pointers loaded from the frame descriptor are generic, which would make every access a FLAT instruction (LDS and
// global sharing both wait counters).  The helpers below pin the address space: LdG/StG = global_load/global_store,
// LdS = ds_read at a byte offset of the block's dynamic LDS.
// =====================================================================================================================
extern __shared__ __align__(16) uint8_t g_dyn_lds[];

#if defined(__HIP_DEVICE_COMPILE__)
template <typename T> __device__ __forceinline__ T LdG(const T* p) {
  return *reinterpret_cast<const __attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p));
}
template <> __device__ __forceinline__ uint4 LdG<uint4>(const uint4* p) {
  typedef uint32_t __attribute__((ext_vector_type(4))) v4;
  const v4 v = *reinterpret_cast<const __attribute__((address_space(1))) v4*>(reinterpret_cast<uintptr_t>(p));
  return make_uint4(v.x, v.y, v.z, v.w);
}
// JXL_NT16 (experiment): the 16-byte accesses of coefficient / pixel planes — data that is touched once per decode — carry the non-temporal
// hint, so that they leave the L2 to what is re-read (entropy-code tables of the SIMT decoders, halo rows of the filter tiles)
#ifdef JXL_NT16
#define JXL_LD16(T, ptr) __builtin_nontemporal_load(ptr)
#else
#define JXL_LD16(T, ptr) (*(ptr))
#endif
template <> __device__ __forceinline__ int4 LdG<int4>(const int4* p) {
  typedef int32_t __attribute__((ext_vector_type(4))) v4;
  const v4 v = JXL_LD16(v4, reinterpret_cast<const __attribute__((address_space(1))) v4*>(reinterpret_cast<uintptr_t>(p)));
  return make_int4(v.x, v.y, v.z, v.w);
}
template <> __device__ __forceinline__ float4 LdG<float4>(const float4* p) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  const v4 v = JXL_LD16(v4, reinterpret_cast<const __attribute__((address_space(1))) v4*>(reinterpret_cast<uintptr_t>(p)));
  return make_float4(v.x, v.y, v.z, v.w);
}
#ifdef JXL_NT16
template <typename T> __device__ __forceinline__ void StG(T* p, T v);
template <> __device__ __forceinline__ void StG<float4>(float4* p, float4 v) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<__attribute__((address_space(1))) v4*>(reinterpret_cast<uintptr_t>(p)));
}
template <> __device__ __forceinline__ void StG<int4>(int4* p, int4 v) {
  typedef int32_t __attribute__((ext_vector_type(4))) v4;
  v4 t; t.x = v.x; t.y = v.y; t.z = v.z; t.w = v.w;
  __builtin_nontemporal_store(t, reinterpret_cast<__attribute__((address_space(1))) v4*>(reinterpret_cast<uintptr_t>(p)));
}
#endif
template <> __device__ __forceinline__ uint2 LdG<uint2>(const uint2* p) {
  typedef uint32_t __attribute__((ext_vector_type(2))) v2;
  const v2 v = *reinterpret_cast<const __attribute__((address_space(1))) v2*>(reinterpret_cast<uintptr_t>(p));
  return make_uint2(v.x, v.y);
}
template <typename T> __device__ __forceinline__ void StG(T* p, T v) {
  *reinterpret_cast<__attribute__((address_space(1))) T*>(reinterpret_cast<uintptr_t>(p)) = v;
}
template <> __device__ __forceinline__ void StG<uint2>(uint2* p, uint2 v) {
  typedef uint32_t __attribute__((ext_vector_type(2))) v2;
  v2 t; t.x = v.x; t.y = v.y;
  *reinterpret_cast<__attribute__((address_space(1))) v2*>(reinterpret_cast<uintptr_t>(p)) = t;
}
#else   // host pass of hipcc only needs the declarations to parse
template <typename T> __device__ __forceinline__ T LdG(const T* p) { return *p; }
template <typename T> __device__ __forceinline__ void StG(T* p, T v) { *p = v; }
#endif
// 16-byte load WITHOUT the non-temporal hint: plane rows a filter tile shares with its neighbours (halo) should stay in the L2 for them
__device__ __forceinline__ float4 LdGKeep(const float4* p) {
  typedef float __attribute__((ext_vector_type(4))) v4;
  const v4 v = *reinterpret_cast<const __attribute__((address_space(1))) v4*>(reinterpret_cast<uintptr_t>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}
template <typename T> __device__ __forceinline__ T LdS(uint32_t byte_off) { return *reinterpret_cast<const T*>(g_dyn_lds + byte_off); }
template <typename T> __device__ __forceinline__ void StS(uint32_t byte_off, T v) { *reinterpret_cast<T*>(g_dyn_lds + byte_off) = v; }
constexpr uint32_t kNotInLds = 0xFFFFFFFFu;

// Bit reader whose next 32-bit word is always already in flight: the refill never waits on global-memory latency.
struct BitReaderP {
  const uint32_t* words;
  uint32_t wpos, wend, nextw;
  uint64_t buf;
  int avail;
  __device__ __forceinline__ uint32_t Load(uint32_t i) const { return i < wend ? LdG(words + i) : 0u; }
  __device__ __forceinline__ void Init(const uint8_t* base, uint64_t bit_pos, uint64_t byte_end) {
    words = reinterpret_cast<const uint32_t*>(base);
    wpos = (uint32_t)(bit_pos >> 5);
    wend = (uint32_t)((byte_end + 3) >> 2);
    buf = (uint64_t)Load(wpos) | ((uint64_t)Load(wpos + 1) << 32);
    wpos += 2;
    nextw = Load(wpos);
    avail = 64;
    const int skip = (int)(bit_pos & 31);
    buf >>= skip; avail -= skip;
    Refill();
  }
  __device__ __forceinline__ void Refill() {
    if (avail <= 32) {
      buf |= (uint64_t)nextw << avail;
      avail += 32;
      wpos++;
      nextw = Load(wpos);
    }
  }
  __device__ __forceinline__ uint32_t Read(int n) {  // n <= 32
    Refill();
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; avail -= n;
    return v;
  }
  __device__ __forceinline__ uint64_t BitPos() const { return (uint64_t)wpos * 32 - (uint64_t)avail; }
};

// Entropy-code tables as seen by the fast decoders: LDS byte offsets when staged, global pointers otherwise.
struct FastCode {
  const uint8_t* ctx_map_g;
  const uint32_t* cfg_g;
  const uint64_t* alias_g;
  uint32_t ctx_map_off, cfg_off, alias_off;   // kNotInLds if the table stayed in global memory
  uint32_t freq_off;                          // StageCodeCompact: per-symbol frequencies (u16) behind the 4-byte alias slots
  uint32_t wide_off, cut_off;                 // StageCode(with_wide): the alias table once more in the form the wave-wide decoder reads (below); kNotInLds if not staged
  uint32_t log_alpha;
  uint32_t cfg_uniform;                       // the hybrid-uint config shared by every cluster, or 0xFFFFFFFF
  __device__ __forceinline__ uint32_t Cluster(uint32_t ctx) const { return ctx_map_off != kNotInLds ? LdS<uint8_t>(ctx_map_off + ctx) : LdG(ctx_map_g + ctx); }
  __device__ __forceinline__ uint32_t Cfg(uint32_t cl) const { return cfg_off != kNotInLds ? LdS<uint32_t>(cfg_off + cl * 4) : LdG(cfg_g + cl); }
  __device__ __forceinline__ uint64_t Alias(uint32_t cluster, uint32_t slot) const {
    if (alias_off == kNotInLds && wide_off == kNotInLds) return LdG(alias_g + (cluster << log_alpha) + slot);
    if (alias_off != kNotInLds) return LdS<uint64_t>(alias_off + (((cluster << log_alpha) + slot) << 3));
    const uint32_t idx = (cluster << log_alpha) + slot;      // (LdAliasAt: defined below)
    const uint2 w = LdS<uint2>(wide_off + (idx << 3));
    const uint32_t cr = LdS<uint16_t>(cut_off + (idx << 1));
    return PackAlias(cr & 0xFFu, cr >> 8, (w.x & 0xFFFu) + 1, (w.y >> 12) & 0xFFFu, (w.y & 0xFFFu) + 1);
  }
};

// alias entry `idx` (= cluster << log_alpha | slot) out of LDS: the plain 8-byte table, or — a code too large for both layouts keeps only the wide one (StageCode) — put
// together again from the wide pair and the {cutoff, aliased symbol} word (the paths that take the channels the wave-wide decoders turn down)
__device__ __forceinline__ uint64_t LdAliasAt(uint32_t alias_off, uint32_t wide_off, uint32_t cut_off, uint32_t idx) {
  if (alias_off != kNotInLds) return LdS<uint64_t>(alias_off + (idx << 3));
  const uint2 w = LdS<uint2>(wide_off + (idx << 3));
  const uint32_t cr = LdS<uint16_t>(cut_off + (idx << 1));
  return PackAlias(cr & 0xFFu, cr >> 8, (w.x & 0xFFFu) + 1, (w.y >> 12) & 0xFFFu, (w.y & 0xFFFu) + 1);
}
__device__ __forceinline__ uint32_t FastSymbol(BitReaderP& br, uint32_t& state, const FastCode& c, uint32_t cluster) {
  const uint32_t la = c.log_alpha;
  const uint32_t res = state & 0xFFF;
  const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
  const uint64_t e = c.Alias(cluster, i);
  const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
  const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
  const bool hit = pos >= cutoff;
  const uint32_t sym = hit ? right : i;
  const uint32_t off = hit ? offs1 + pos : pos;
  const uint32_t freq = hit ? freq1 : freq0;
  state = freq * (state >> 12) + off;
  if (state < (1u << 16)) state = (state << 16) | br.Read(16);
  return sym;
}
__device__ __forceinline__ uint32_t FastHybrid(BitReaderP& br, uint32_t& state, const FastCode& c, uint32_t cluster) {
  const uint32_t cfg = c.Cfg(cluster);
  uint32_t tok = FastSymbol(br, state, c, cluster);
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  const uint32_t split = 1u << split_exp;
  if (tok < split) return tok;
  uint32_t nbits = split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb));
  nbits &= 31;
  const uint32_t low = tok & ((1u << lsb) - 1);
  tok >>= lsb;
  const uint32_t bits = nbits ? br.Read((int)nbits) : 0;
  const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
  return (((hi << nbits) | bits) << lsb) | low;
}

// Compile-time selection of the table location (ALL_LDS = every table of the code was staged): keeps vmcnt waits out
// of the LDS variant.
template <bool ALL_LDS> __device__ __forceinline__ uint32_t ClusterT(const FastCode& c, uint32_t ctx) {
  if (ALL_LDS) return LdS<uint8_t>(c.ctx_map_off + ctx);
  return c.Cluster(ctx);
}
template <bool ALL_LDS> __device__ __forceinline__ uint32_t FastHybridT(BitReaderP& br, uint32_t& state, const FastCode& c, uint32_t cluster) {
  if (!ALL_LDS) return FastHybrid(br, state, c, cluster);
  const uint32_t la = c.log_alpha;
  const uint32_t cfg = LdS<uint32_t>(c.cfg_off + cluster * 4);
  const uint32_t res = state & 0xFFF;
  const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
  const uint64_t e = LdS<uint64_t>(c.alias_off + (((cluster << la) + i) << 3));
  const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
  const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
  const bool hit = pos >= cutoff;
  uint32_t tok = hit ? right : i;
  const uint32_t off = hit ? offs1 + pos : pos;
  const uint32_t freq = hit ? freq1 : freq0;
  state = freq * (state >> 12) + off;
  // (the caller refilled: at least 33 bits are buffered, 17 after the renormalisation — enough for most extra-bit fields)
  if (state < (1u << 16)) { state = (state << 16) | (uint32_t)(br.buf & 0xFFFFu); br.buf >>= 16; br.avail -= 16; }
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  const uint32_t split = 1u << split_exp;
  if (tok < split) return tok;
  uint32_t nbits = split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb));
  nbits &= 31;
  const uint32_t low = tok & ((1u << lsb) - 1);
  tok >>= lsb;
  if ((int)nbits > br.avail) br.Refill();
  const uint32_t bits = (uint32_t)(br.buf & ((1ull << nbits) - 1));
  br.buf >>= nbits; br.avail -= (int)nbits;
  const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
  return (((hi << nbits) | bits) << lsb) | low;
}

// LDS-only variants used by the serial fast path: no vector-memory instruction, hence no vmcnt wait, in the token loop.
struct BitReaderW {      // reads 32-bit words from an LDS window at win_off holding words win_base.. of the stream
  uint32_t wpos, win_base, win_off;
  uint64_t buf;
  int avail;
  __device__ __forceinline__ void Refill() {
    if (avail <= 32) {
      buf |= (uint64_t)LdS<uint32_t>(win_off + ((wpos - win_base) << 2)) << avail;
      avail += 32;
      wpos++;
    }
  }
  __device__ __forceinline__ uint32_t Read(int n) {
    Refill();
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; avail -= n;
    return v;
  }
  __device__ __forceinline__ uint64_t BitPos() const { return (uint64_t)wpos * 32 - (uint64_t)avail; }
};
// Cooperative copy of an entropy code into LDS (all threads of the block) starting at byte offset `base`; tables
// that do not fit in [base, base + budget) stay in global memory.  Returns the bytes used.
// with_wide: when budget is left, the alias table is staged a second time in the layout of the wave-wide decoder (DecodeChannelWave): per slot one 64-bit
// pair of ready-made candidates — low word: the slot's own symbol, high word: the aliased one, each {frequency - 1 [0:12), offset [12:24), value [24:32)} where
// value is the symbol's UnpackSigned() as a signed byte when the symbol is a complete hybrid-uint token of its cluster's configuration, -128 (kWideEscape)
// when extra bits follow — and one u16 {cutoff [0:8), aliased symbol [8:16)}.  10 bytes per slot instead of 8.
constexpr int32_t kWideEscape = -128;
__device__ __forceinline__ uint32_t WideValue(uint32_t tok, uint32_t cfg) {
  const uint32_t split = 1u << (cfg & 0xFF);
  return (tok < split && tok < 255u) ? ((uint32_t)UnpackSigned(tok) & 0xFFu) : ((uint32_t)kWideEscape & 0xFFu);
}
__device__ uint32_t StageCode(const DevCode& g, FastCode& fc, uint32_t base, uint32_t budget, bool with_ctx_map, bool with_wide = false, bool with_plain = true) {
  uint32_t used = 0;
  fc.log_alpha = g.log_alpha;
  fc.ctx_map_g = g.ctx_map; fc.cfg_g = g.cfg; fc.alias_g = g.alias;
  fc.ctx_map_off = fc.cfg_off = fc.alias_off = kNotInLds;
  fc.wide_off = fc.cut_off = kNotInLds;
  {
    const uint32_t c0 = LdG(g.cfg);
    bool same = true;
    for (uint32_t i = threadIdx.x; i < g.num_clusters; i += blockDim.x) same = same && LdG(g.cfg + i) == c0;
    fc.cfg_uniform = __syncthreads_and(same) ? c0 : 0xFFFFFFFFu;
  }
  const uint32_t cfg_bytes = (g.num_clusters * 4 + 15) & ~15u;
  if (used + cfg_bytes <= budget) {
    for (uint32_t i = threadIdx.x; i < g.num_clusters; i += blockDim.x) StS<uint32_t>(base + used + i * 4, LdG(g.cfg + i));
    fc.cfg_off = base + used; used += cfg_bytes;
  }
  if (with_ctx_map) {
    const uint32_t n = (g.num_ctx + 15) & ~15u;
    if (used + n <= budget) {
      for (uint32_t i = threadIdx.x; i < g.num_ctx; i += blockDim.x) StS<uint8_t>(base + used + i, LdG(g.ctx_map + i));
      fc.ctx_map_off = base + used; used += n;
    }
  }
  const uint32_t n_alias = g.num_clusters << g.log_alpha;
  const uint32_t wide_bytes = n_alias * 8 + ((n_alias * 2 + 15) & ~15u);
  // both layouts when they fit; the wide one alone when only it does and the caller reads it (its readers are the fast ones; the others put the plain entry together
  // again: LdAliasAt); else the plain one as before
  const bool wide_ok = with_wide && fc.cfg_off != kNotInLds && used + wide_bytes <= budget;
  if (with_plain && used + n_alias * 8 + (wide_ok ? wide_bytes : 0) <= budget) {
    for (uint32_t i = threadIdx.x; i < n_alias; i += blockDim.x) StS<uint64_t>(base + used + i * 8, LdG(g.alias + i));
    fc.alias_off = base + used; used += n_alias * 8;
  } else if (with_plain && !wide_ok && used + n_alias * 8 <= budget) {
    for (uint32_t i = threadIdx.x; i < n_alias; i += blockDim.x) StS<uint64_t>(base + used + i * 8, LdG(g.alias + i));
    fc.alias_off = base + used; used += n_alias * 8;
  }
  {
    if (with_wide && fc.cfg_off != kNotInLds && used + wide_bytes <= budget) {
      const uint32_t wo = base + used, co = wo + n_alias * 8;
      for (uint32_t i = threadIdx.x; i < n_alias; i += blockDim.x) {
        const uint64_t e = LdG(g.alias + i);
        const uint32_t cfg = LdG(g.cfg + (i >> g.log_alpha)), slot = i & ((1u << g.log_alpha) - 1);
        const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
        const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
        const uint32_t lo = (max(freq0, 1u) - 1) | (WideValue(slot, cfg) << 24);
        const uint32_t hi = (max(freq1, 1u) - 1) | ((offs1 & 0xFFFu) << 12) | (WideValue(right, cfg) << 24);
        StS<uint64_t>(wo + i * 8, (uint64_t)lo | ((uint64_t)hi << 32));
        StS<uint16_t>(co + i * 2, (uint16_t)(cutoff | (right << 8)));
      }
      fc.wide_off = wo; fc.cut_off = co; used += wide_bytes;
    }
  }
  return used;
}

// The same for the SIMT HF kernel's all-in-LDS instantiations, with the alias table in 6 instead of 8 bytes per slot: a 4-byte slot
// cutoff[0:8) right[8:16) offs1[16:29) and one u16 frequency per symbol — freq0 of slot i is the frequency of symbol i, freq1 that of symbol
// `right` (host_parse.cc BuildAlias), so the reader looks the frequency up by the symbol it decoded (one more dependent LDS read per token; 57 -> 43 KB
// for the bench's AC code: two more pixel workgroups beside an HF workgroup on every CU).  All or nothing: returns 0 and leaves alias_off = kNotInLds
// if [base, base + budget) is too small.
__device__ uint32_t StageCodeCompact(const DevCode& g, FastCode& fc, uint32_t base, uint32_t budget) {
  fc.log_alpha = g.log_alpha;
  fc.ctx_map_g = g.ctx_map; fc.cfg_g = g.cfg; fc.alias_g = g.alias;
  fc.ctx_map_off = fc.cfg_off = fc.alias_off = fc.freq_off = fc.wide_off = fc.cut_off = kNotInLds;
  {
    const uint32_t c0 = LdG(g.cfg);
    bool same = true;
    for (uint32_t i = threadIdx.x; i < g.num_clusters; i += blockDim.x) same = same && LdG(g.cfg + i) == c0;
    fc.cfg_uniform = __syncthreads_and(same) ? c0 : 0xFFFFFFFFu;
  }
  const uint32_t cfg_bytes = (g.num_clusters * 4 + 15) & ~15u, map_bytes = (g.num_ctx + 15) & ~15u;
  const uint32_t n_alias = g.num_clusters << g.log_alpha;
  const uint32_t total = cfg_bytes + map_bytes + ((n_alias * 6 + 15) & ~15u);
  if (total > budget) return 0;
  for (uint32_t i = threadIdx.x; i < g.num_clusters; i += blockDim.x) StS<uint32_t>(base + i * 4, LdG(g.cfg + i));
  fc.cfg_off = base;
  for (uint32_t i = threadIdx.x; i < g.num_ctx; i += blockDim.x) StS<uint8_t>(base + cfg_bytes + i, LdG(g.ctx_map + i));
  fc.ctx_map_off = base + cfg_bytes;
  const uint32_t slots = base + cfg_bytes + map_bytes, freqs = slots + n_alias * 4;
  for (uint32_t i = threadIdx.x; i < n_alias; i += blockDim.x) {
    const uint64_t e = LdG(g.alias + i);
    StS<uint32_t>(slots + i * 4, (uint32_t)(e & 0xFFFF) | ((uint32_t)((e >> 29) & 0x1FFF) << 16));
    StS<uint16_t>(freqs + i * 2, (uint16_t)((e >> 16) & 0x1FFF));            // freq0: symbol i of this cluster
  }
  __syncthreads();
  // freq1 belongs to symbol `right` (equal to what the first pass wrote there, except in a one-symbol histogram, whose slots carry freq0 = 0)
  for (uint32_t i = threadIdx.x; i < n_alias; i += blockDim.x) {
    const uint64_t e = LdG(g.alias + i);
    const uint32_t right = (uint32_t)((e >> 8) & 0xFF), cluster_base = i & ~((1u << g.log_alpha) - 1);
    if (right < (1u << g.log_alpha)) StS<uint16_t>(freqs + (cluster_base + right) * 2, (uint16_t)((e >> 42) & 0x1FFF));
  }
  fc.alias_off = slots; fc.freq_off = freqs;
  return total;
}

// ---- Modular channel decode, cooperative version -------------------------------------------------------------------------
// One wavefront decodes one sub-stream: lane 0 runs the serial chain, the other 63 lanes build LUTs and move data.
// Several wavefronts (= several LF groups of one frame) share a workgroup and therefore one LDS copy of the MA tree and
// the entropy code; everything else is per wavefront at `wb` and synchronised with wave-level fences only.
constexpr int kLdsTreeMax = 1024;   // nodes copied to LDS (16 KiB); larger trees are walked in global memory
// per-wavefront LDS region
constexpr uint32_t kLutOff = 0, kWorkOff = 2048;
constexpr uint32_t kWinOff = 3072, kWinWords = 528;                   // bit-stream window: 2112 B (>= 256 samples x 48 bits + slack)
constexpr uint32_t kRowOff = kWinOff + kWinWords * 4, kRowMax = 256;  // two sample rows (current / previous) of up to 256 ints
constexpr uint32_t kChunkOff = kRowOff + 2 * kRowMax * 4;             // 256-sample staging buffer for wider channels
constexpr uint32_t kWaveLds = 8448;                                   // >= kChunkOff + 1024 and >= the 8 KiB placement bitmap
static_assert(kChunkOff + 1024 <= kWaveLds, "per-wave LDS layout");
#ifndef JXL_LF_MINW
#define JXL_LF_MINW 2      // LfDecodeKernel<true>: VGPR budget 512 / JXL_LF_MINW per lane (256: no spills.  Rounds 3-5 ran it at 3 = 170 registers with 210 spilled, for the wavefronts of other stages
                           // beside it; since the SIMT kernel took over the LF streams of resident pipelines this one runs at cold starts and for handed-back streams, and measured the same either way)
#endif
#ifndef JXL_IDCT_T4
#define JXL_IDCT_T4 128   // threads of an IdctTileKernel<4> workgroup (A/B knob: 256 = four wavefronts share the 14 KB tile)
#endif
#ifndef JXL_IDCT_MINW
#define JXL_IDCT_MINW 5    // IdctTileKernel<4>: five wavefronts per SIMD = 96 VGPRs (24 dwords of scratch in the SPECIAL variant): four of them fit beside an HF wavefront (80 VGPRs) where 110 registers allowed three (r04: steady state 71.0 -> 69.7 ms per step, alone 19.2 -> 18.8)
#endif
// The dynamic-LDS ceiling of a kernel (static __shared__ arrays count against the same 160 KB): a refused request must not pass silently — launches above the
// default 64 KB would then be refused one by one, each leaving an error code for some later runtime call to report.
static void SetMaxDynamicLds(const void* func, int bytes, const char* name) {
  const hipError_t e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e != hipSuccess) { (void)hipGetLastError(); fprintf(stderr, "[jxl-hip] cannot raise the dynamic LDS limit of %s to %d bytes: %s\n", name, bytes, hipGetErrorString(e)); }
}
constexpr uint32_t kLfWaves = 4;                                         // wavefronts per workgroup of the Modular group kernels
constexpr uint32_t kLfDecWaves = 2, kLfDecGroups = 4;                    // LfDecodeKernel: two wavefronts share four LF groups
// the shared part of the LDS (tree copy or per-wavefront pruned slices, then the entropy code) follows the per-wavefront regions

__device__ __forceinline__ void WaveSync() {   // orders LDS/global accesses between the lanes of ONE wavefront
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}


// Weighted-predictor state (context_predict.h weighted::State) held in LDS: same arithmetic as WPState (jxl_dev.h), the
// five error arrays (2 rows of xsize + 2 ints each) at a byte offset of the dynamic LDS instead of global scratch.
struct WPStateLds {
  uint32_t base;            // byte offset; layout [error | pe0 | pe1 | pe2 | pe3], each 2 * (xsize + 2) ints
  uint32_t div_off;         // 64-entry table of (1 << 24) / (i + 1) (context_predict.h kDivLookup) instead of a division per use
  int32_t xsize;
  bool narrow;              // 32-bit intermediates are exact for this image (samples of at most 12 bits)
  int64_t prediction_store[4];
  int64_t pred;
  template <typename WI> static __device__ __forceinline__ WI AbsT(WI v) { return v < 0 ? -v : v; }
  __device__ __forceinline__ uint32_t Arr(int a) const { return base + (uint32_t)a * (uint32_t)(xsize + 2) * 8u; }
  __device__ __forceinline__ int32_t Ld(int a, int32_t i) const { return LdS<int32_t>(Arr(a) + (uint32_t)i * 4u); }
  __device__ __forceinline__ void St(int a, int32_t i, int32_t v) const { StS<int32_t>(Arr(a) + (uint32_t)i * 4u, v); }
  static constexpr uint32_t Bytes(int32_t xs) { return 5u * 2u * (uint32_t)(xs + 2) * 4u; }
  __device__ __forceinline__ void Init(uint32_t base_off, uint32_t div_table_off, int32_t xs, uint32_t lane, bool narrow_ok) {   // all lanes of the wavefront
    base = base_off; xsize = xs; div_off = div_table_off; narrow = narrow_ok;
    for (uint32_t i = lane; i < Bytes(xs) / 4; i += 64) StS<int32_t>(base + i * 4, 0);
    StS<uint32_t>(div_off + lane * 4, (1u << 24) / (lane + 1));
  }
  __device__ __forceinline__ uint32_t Div(uint32_t i) const { return LdS<uint32_t>(div_off + i * 4); }
  __device__ __forceinline__ uint32_t ErrorWeight(uint64_t x, uint32_t maxweight) const {
    int shift = FloorLog2u64(x + 1) - 5;
    if (shift < 0) shift = 0;
    return 4 + ((maxweight * Div((uint32_t)(x >> shift))) >> shift);
  }
  // WI = int64_t (reference arithmetic) or int32_t: identical results whenever no intermediate exceeds 31 bits, which holds
  // for samples of at most 12 bits (values * 8, errors and the weighted sums stay below 2^28); only the final product with
  // the division table needs 64 bits.
  template <typename WI> __device__ __forceinline__ int64_t PredictT(const WPHeader& hdr, int x, int y, WI N, WI W, WI NE, WI NW, WI NN, int32_t* max_err) {
    WI* prediction = reinterpret_cast<WI*>(prediction_store);
    const int32_t cur_row = (y & 1) ? 0 : (xsize + 2);
    const int32_t prev_row = (y & 1) ? (xsize + 2) : 0;
    const int32_t pos_N = prev_row + x;
    const int32_t pos_NE = x < xsize - 1 ? pos_N + 1 : pos_N;
    const int32_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
    // all sixteen state reads are issued before anything waits for one of them (one LDS round trip instead of eight)
    uint32_t e0[4], e1[4], e2[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { e0[i] = (uint32_t)Ld(1 + i, pos_N); e1[i] = (uint32_t)Ld(1 + i, pos_NE); e2[i] = (uint32_t)Ld(1 + i, pos_NW); }
    const WI teW = x == 0 ? 0 : Ld(0, cur_row + x - 1);
    const WI teN = Ld(0, pos_N);
    const WI teNW = Ld(0, pos_NW);
    const WI teNE = Ld(0, pos_NE);
    // error weights: the four division-table reads likewise go out together
    int shift[4];
    uint32_t quot[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const uint64_t sum = (uint64_t)e0[i] + e1[i] + e2[i];
      shift[i] = FloorLog2u64(sum + 1) - 5;
      if (shift[i] < 0) shift[i] = 0;
      quot[i] = Div((uint32_t)(sum >> shift[i]));
    }
    uint32_t weights[4];
#pragma unroll
    for (int i = 0; i < 4; i++) weights[i] = 4 + (((uint32_t)hdr.w[i] * quot[i]) >> shift[i]);
    N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
    const WI sumWN = teN + teW;
    WI p = teW;
    if (AbsT(teN) > AbsT(p)) p = teN;
    if (AbsT(teNW) > AbsT(p)) p = teNW;
    if (AbsT(teNE) > AbsT(p)) p = teNE;
    *max_err = (int32_t)p;
    prediction[0] = W + NE - N;
    prediction[1] = N - (((sumWN + teNE) * hdr.p1) >> 5);
    prediction[2] = W - (((sumWN + teNW) * hdr.p2) >> 5);
    prediction[3] = N - ((teNW * hdr.p3[0] + teN * hdr.p3[1] + teNE * hdr.p3[2] + (NN - N) * hdr.p3[3] + (NW - W) * hdr.p3[4]) >> 5);
    uint32_t wsum = weights[0] + weights[1] + weights[2] + weights[3];
    const int lw = FloorLog2u64(wsum);
    wsum = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) { weights[i] >>= (lw - 4); wsum += weights[i]; }
    const uint32_t inv = Div(wsum - 1);
    WI sum = (WI)(wsum >> 1) - 1;
#pragma unroll
    for (int i = 0; i < 4; i++) sum += prediction[i] * (WI)weights[i];
    pred = ((int64_t)sum * (int64_t)inv) >> 24;
    if (((teN ^ teW) | (teN ^ teNW)) > 0) return pred;
    const int64_t mx = Max64(W, Max64(NE, N)), mn = Min64(W, Min64(NE, N));
    pred = Max64(mn, Min64(mx, pred));
    return pred;
  }
  __device__ __forceinline__ int64_t Predict(const WPHeader& hdr, int x, int y, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int32_t* max_err) {
    if (narrow) return PredictT<int32_t>(hdr, x, y, (int32_t)N, (int32_t)W, (int32_t)NE, (int32_t)NW, (int32_t)NN, max_err);
    return PredictT<int64_t>(hdr, x, y, N, W, NE, NW, NN, max_err);
  }
  template <typename WI> __device__ __forceinline__ void UpdateT(WI val, int x, int y) {
    const WI* prediction = reinterpret_cast<const WI*>(prediction_store);
    const int32_t cur_row = (y & 1) ? 0 : (xsize + 2);
    const int32_t prev_row = (y & 1) ? (xsize + 2) : 0;
    val *= 8;
    int32_t acc[4];
#pragma unroll
    for (int i = 0; i < 4; i++) acc[i] = Ld(1 + i, prev_row + x + 1);       // (read together, ahead of the stores)
    St(0, cur_row + x, (int32_t)((WI)pred - val));
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int32_t err = (int32_t)((AbsT(prediction[i] - val) + 3) >> 3);
      St(1 + i, cur_row + x, err);
      St(1 + i, prev_row + x + 1, acc[i] + err);
    }
  }
  // The 32-bit variant is only exact while every sample seen so far is small.  The image's bit depth promises that for plain
  // streams, but residual channels (RCT, squeeze) or a hostile stream may exceed it: the first sample outside +-4095 switches
  // this channel to the 64-bit arithmetic for good (the state arrays are the same in both variants; all inputs of the
  // predictions made so far were within the bound).
  __device__ __forceinline__ void UpdateStores(int64_t val, int x, int y) { if (narrow) UpdateT<int32_t>((int32_t)val, x, y); else UpdateT<int64_t>(val, x, y); }
  __device__ __forceinline__ void NoteSample(int64_t val) { if (narrow && (uint64_t)(val + 4095) > 8190ull) narrow = false; }
  __device__ __forceinline__ void Update(int64_t val, int x, int y) { UpdateStores(val, x, y); NoteSample(val); }
};
constexpr int32_t kWpLdsMaxW = 256;                                   // channel widths whose WP state fits the per-wavefront LDS slot
constexpr uint32_t kWpLdsBytes = WPStateLds::Bytes(kWpLdsMaxW) + 256;   // 10 320 B of error rows + the 64-entry division table

struct ModTables {
  const TreeNode* tree_g;
  bool tree_in_lds;         // LDS copy at node_base (leaves rewritten: a = predictor | cluster << 8)
  uint32_t tree_cap;
  uint32_t node_base;       // LDS byte offset of the node array Node() reads when tree_in_lds
  uint32_t prune_off, prune_cap;   // this wavefront's slice of the tree region for a pruned per-(stream, channel) subtree of a
                                   // tree too large to copy whole (prune_cap nodes; 0 = none)
  const uint8_t* ctx_map_g;
  uint32_t wb;              // base of this wavefront's private LDS region
  uint32_t wp_off;          // this wavefront's LDS slot for the weighted-predictor state (kWpLdsBytes), or 0xFFFFFFFF: global scratch
  FastCode code;
  __device__ __forceinline__ TreeNode Node(uint32_t i) const {
    if (tree_in_lds) {
      const uint4 v = LdS<uint4>(node_base + i * 16);
      return TreeNode{(int32_t)v.x, (int32_t)v.y, v.z, v.w};
    }
    const uint4 v = LdG(reinterpret_cast<const uint4*>(tree_g + i));
    return TreeNode{(int32_t)v.x, (int32_t)v.y, v.z, v.w};
  }
};

__device__ __forceinline__ int32_t PropValue(int p, int chan, uint32_t stream_id, int x, int y, int32_t W, int32_t N, int32_t NW, int32_t NE, int32_t NN, int32_t WW,
                                             int32_t prev9, int32_t wp_err) {
  switch (p) {
    case 0: return chan;
    case 1: return (int32_t)stream_id;
    case 2: return y;
    case 3: return x;
    case 4: return N < 0 ? (int32_t)(0u - (uint32_t)N) : N;
    case 5: return W < 0 ? (int32_t)(0u - (uint32_t)W) : W;
    case 6: return N;
    case 7: return W;
    case 8: return (int32_t)((uint32_t)W - (uint32_t)prev9);
    case 9: return (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
    case 10: return (int32_t)((uint32_t)W - (uint32_t)NW);
    case 11: return (int32_t)((uint32_t)NW - (uint32_t)N);
    case 12: return (int32_t)((uint32_t)N - (uint32_t)NE);
    case 13: return (int32_t)((uint32_t)N - (uint32_t)NN);
    case 14: return (int32_t)((uint32_t)W - (uint32_t)WW);
    default: return wp_err;
  }
}

// Prediction in 32-bit wrap-around arithmetic where that is exact (the final sample is truncated to int32 anyway and
// the clamped gradient only uses N+W-NW when it lies between N and W), 64-bit for the averaging predictors.
__device__ __forceinline__ int32_t Predict(uint32_t predictor, int32_t W, int32_t N, int32_t NW, int32_t NE, int32_t NN, int32_t WW, int32_t NEE, int64_t wp_pred) {
  switch (predictor) {
    case 0: return 0;
    case 1: return W;
    case 2: return N;
    case 3: return (int32_t)(((int64_t)W + N) / 2);
    case 4: { const int64_t pp = (int64_t)W + N - NW; return Abs64(pp - W) < Abs64(pp - N) ? W : N; }
    case 5: { const int32_t m = min(N, W), M = max(N, W); return NW < m ? M : (NW > M ? m : (int32_t)((uint32_t)N + (uint32_t)W - (uint32_t)NW)); }
    case 6: return (int32_t)((wp_pred + 3) >> 3);
    case 7: return NE;
    case 8: return NW;
    case 9: return WW;
    case 10: return (int32_t)(((int64_t)W + NW) / 2);
    case 11: return (int32_t)(((int64_t)N + NW) / 2);
    case 12: return (int32_t)(((int64_t)N + NE) / 2);
    default: return (int32_t)((6 * (int64_t)N - 2 * (int64_t)NN + 7 * (int64_t)W + WW + NEE + 3 * (int64_t)NE + 8) / 16);
  }
}

// Serial inner loop of the LDS-only fast path, specialised at compile time:
//   ROWMODE 0: first row (N = NW = W), 1: previous row needed (read from LDS, next value prefetched), 2: W-only rows
//   PROP9: context from W+N-NW through the LUT (else one cluster per row);  UPRED: 0 zero, 1 W, 5 clamped gradient
// cluster of property value v under the LDS copy of the (single-property) subtree at `pos` (leaves: a = predictor | cluster << 8)
__device__ __forceinline__ uint32_t WalkCluster(uint32_t node_base, uint32_t pos, int32_t v) {
  uint4 n = LdS<uint4>(node_base + pos * 16);
  while ((int32_t)n.x >= 0) { pos = v > (int32_t)n.y ? n.z : n.w; n = LdS<uint4>(node_base + pos * 16); }
  return n.z >> 8;
}
struct ChunkState { BitReaderW bw; uint32_t state; int32_t left, nw; };
__device__ __forceinline__ uint32_t Uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
template <int ROWMODE, bool PROP9, int UPRED, bool UCFG>
__device__ __forceinline__ void DecodeChunkLds(ChunkState& st, int x0, int x1, uint32_t prev, uint32_t obase, uint32_t lut_off, uint32_t first_off, uint32_t cl_row,
                                               uint32_t cfg_off, uint32_t cfg_uniform, uint32_t alias_off, uint32_t la, uint32_t wide_subroot, uint32_t node_base) {
  // loop invariants into scalar registers (they come out of LDS-resident tables, i.e. vector registers)
  wide_subroot = Uniform(wide_subroot); node_base = Uniform(node_base);
  prev = Uniform(prev); obase = Uniform(obase); lut_off = Uniform(lut_off); first_off = Uniform(first_off); cl_row = Uniform(cl_row);
  cfg_off = Uniform(cfg_off); cfg_uniform = Uniform(cfg_uniform); alias_off = Uniform(alias_off); la = Uniform(la);
  x0 = (int)Uniform((uint32_t)x0); x1 = (int)Uniform((uint32_t)x1);
  BitReaderW bw = st.bw;
  uint32_t state = st.state;
  int32_t left = st.left, nw = st.nw;
  int32_t n_next = ROWMODE == 1 ? LdS<int32_t>(prev + 4 * x0) : 0;
  for (int x = x0; x < x1; x++) {
    int32_t W, N, NW;
    if (ROWMODE == 0) { W = x ? left : 0; N = W; NW = W; }
    else if (ROWMODE == 1) {
      N = n_next;
      n_next = LdS<int32_t>(prev + 4 * (x + 1));     // one slot of slack exists past the row (kRowMax + chunk buffers follow)
      W = x ? left : N; NW = x ? nw : W; nw = N;
    } else { W = x ? left : LdS<int32_t>(first_off); N = W; NW = W; }
    uint32_t cluster = cl_row;
    if (PROP9) {
      const int32_t v0 = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
      const int32_t v = v0 < -512 ? -512 : (v0 > 511 ? 511 : v0);
      cluster = LdS<uint16_t>(lut_off + 2 * (uint32_t)(v + 512));
      if (__builtin_expect(wide_subroot != 0xFFFFFFFFu && v != v0, 0)) cluster = WalkCluster(node_base, wide_subroot, v0);   // rare: beyond the LUT
    }
    int32_t guess;
    if (UPRED == 0) guess = 0;
    else if (UPRED == 1) guess = W;
    else { const int32_t m = min(N, W), M = max(N, W); guess = NW < m ? M : (NW > M ? m : (int32_t)((uint32_t)N + (uint32_t)W - (uint32_t)NW)); }
    // --- ANS symbol + hybrid integer (dec_ans.h), tables in LDS
    const uint32_t res = state & 0xFFF;
    const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
    const uint64_t e = LdS<uint64_t>(alias_off + (((cluster << la) + i) << 3));
    const uint32_t cfg = UCFG ? cfg_uniform : LdS<uint32_t>(cfg_off + cluster * 4);
    const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
    const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
    const bool hit = pos >= cutoff;
    uint32_t tok = hit ? right : i;
    const uint32_t off = hit ? offs1 + pos : pos;
    const uint32_t freq = hit ? freq1 : freq0;
    state = freq * (state >> 12) + off;
    if (state < (1u << 16)) state = (state << 16) | bw.Read(16);
    const uint32_t split_exp = cfg & 0xFF;
    const uint32_t split = 1u << split_exp;
    if (tok >= split) {
      const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
      uint32_t nbits = split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb));
      nbits &= 31;
      const uint32_t low = tok & ((1u << lsb) - 1);
      tok >>= lsb;
      const uint32_t bits = nbits ? bw.Read((int)nbits) : 0;
      const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
      tok = (((hi << nbits) | bits) << lsb) | low;
    }
    const int32_t val = (int32_t)((uint32_t)UnpackSigned(tok) + (uint32_t)guess);
    StS<int32_t>(obase + 4 * x, val);
    if (x == 0) StS<int32_t>(first_off, val);   // W of the next row's first sample
    left = val;
  }
  st.bw = bw; st.state = state; st.left = left; st.nw = nw;
}

// All 64 lanes of the wavefront call this.  Lane 0 decodes; the others help with LUT, bit-stream window and row I/O.
// Semantics identical to DecodeModularChannel (jxl_dev.h).
// ---- "ballot" decode of a channel under a general MA tree: the whole wavefront decodes the channel together.  Every value of the
// serial chain (neighbours, ANS state, bit buffer, weighted-predictor arithmetic) is computed redundantly by all 64 lanes — that
// costs nothing, a wavefront instruction takes the same time for one active lane as for 64 — and the part that used to be a
// pointer chase through LDS is spread over the lanes: lane j owns the inner nodes j, 64 + j, ... of the channel's subtree (R
// register rows, at most 64 R inner nodes and leaves after the static splits are resolved), evaluates their properties and
// comparisons, R ballots yield all decisions of the tree as 64-bit scalars, and the walk from the root is scalar bit tests with
// the child tables read across lanes (v_readlane) — no LDS round trip per tree level (the single-lane loop pays two).
// R == 1: the lane's property comes out of a select chain over the properties the subtree uses; R > 1: lane 0 puts the 16
// property values into LDS and every lane fetches those of its nodes (one LDS round trip for the whole tree).  Channels whose
// subtree never looks at the weighted predictor (no predictor 6 leaf, no property 15 split) skip its arithmetic altogether.
constexpr uint32_t kBallotMax = 512;
struct BallotArgs { uint32_t qi_off, ql_off, pair_off, ni, nl, root_code; bool use_wp; };
template <int R>
__device__ __forceinline__ void DecodeRowsBallot(BitReaderP& br, uint32_t& state, const ModTables& T, const ModularCtx& mc, const ChannelDesc& ch, int chan, WPStateLds& wpl, const BallotArgs& ba) {
  const uint32_t lane = threadIdx.x & 63, wb = T.wb;
  const uint32_t root_code = ba.root_code;
  int32_t my_prop[R], my_val[R], leaf_val[R];
  uint32_t my_pair[R], leaf_a[R], leaf_b[R];
  uint32_t used = 0;
  bool wp_leaf = false;
#pragma unroll
  for (int r = 0; r < R; r++) {
    const uint32_t idx = (uint32_t)r * 64 + lane;
    my_prop[r] = R == 1 ? -1 : 0; my_val[r] = 0x7FFFFFFF; my_pair[r] = 0; leaf_a[r] = 0; leaf_b[r] = 1; leaf_val[r] = 0;
    if (idx < ba.ni) { const TreeNode n = T.Node(LdS<uint16_t>(ba.qi_off + 2 * idx)); my_prop[r] = n.prop & 15; my_val[r] = n.val; my_pair[r] = LdS<uint32_t>(ba.pair_off + 4 * idx); }
    if (idx < ba.nl) { const TreeNode n = T.Node(LdS<uint16_t>(ba.ql_off + 2 * idx)); leaf_a[r] = n.a; leaf_b[r] = n.b; leaf_val[r] = n.val; }
    for (int k = 0; k < 16; k++) if (__ballot(idx < ba.ni && my_prop[r] == k)) used |= 1u << k;
    wp_leaf |= __ballot(idx < ba.nl && (leaf_a[r] & 0xFF) == 6) != 0;
  }
  used = Uniform(used);
  const bool wp_here = ba.use_wp && (((used >> 15) & 1) || wp_leaf);
  WaveSync();     // the tables have been read: the window region may be overwritten
  const int w = ch.w, h = ch.h;
  const uint32_t cfg_off = T.code.cfg_off, alias_off = T.code.alias_off, la = T.code.log_alpha;
  const uint32_t wend = br.wend;
  const uint32_t pv = wb + kWorkOff + 896;
  // the incoming ANS state and bit position live in lane 0
  state = Uniform(state);
  const uint64_t bp = br.BitPos();
  const uint64_t bp0 = ((uint64_t)Uniform((uint32_t)(bp >> 32)) << 32) | Uniform((uint32_t)bp);
  BitReaderW bw;
  bw.wpos = (uint32_t)(bp0 >> 5); bw.win_base = 0; bw.buf = 0; bw.avail = 0; bw.win_off = wb + kWinOff;
  uint32_t skip_bits = (uint32_t)(bp0 & 31);
  uint32_t cur = wb + kRowOff, prev = wb + kRowOff + kRowMax * 4, prev2 = wb + kChunkOff;   // three row buffers, rotated
  for (int y = 0; y < h; y++) {
    int32_t* p = ch.data + (size_t)y * ch.stride;
    const uint32_t wbase = bw.wpos;
    WaveSync();    // (the previous row's window reads are done)
    for (uint32_t i = lane; i < kWinWords; i += 64) StS<uint32_t>(wb + kWinOff + i * 4, wbase + i < wend ? LdG(br.words + wbase + i) : 0u);
    WaveSync();
    bw.win_base = wbase;
    if (skip_bits != 0xFFFFFFFFu) { bw.buf = 0; bw.avail = 0; bw.Refill(); bw.buf >>= skip_bits; bw.avail -= (int)skip_bits; skip_bits = 0xFFFFFFFFu; }
    int32_t left = 0, left2 = 0, prev9 = 0;
    int32_t up0 = 0, up1 = 0, up2 = 0, up3 = 0;
    if (y > 0) { up1 = LdS<int32_t>(prev); up2 = w > 1 ? LdS<int32_t>(prev + 4) : 0; up3 = w > 2 ? LdS<int32_t>(prev + 8) : 0; }
    for (int x = 0; x < w; x++) {
      const int32_t up4 = (y > 0 && x + 3 < w) ? LdS<int32_t>(prev + 4 * (x + 3)) : 0;
      const int32_t W = x ? left : (y ? up1 : 0);
      const int32_t N = y ? up1 : W;
      const int32_t NW = (x && y) ? up0 : W;
      const int32_t NE = (x + 1 < w && y) ? up2 : N;
      const int32_t WW = x > 1 ? left2 : W;
      const int32_t NN = y > 1 ? LdS<int32_t>(prev2 + 4 * x) : N;
      const int32_t NEE = (x + 2 < w && y) ? up3 : NE;
      int64_t wp_pred = 0;
      int32_t wp_err = 0;
      if (wp_here) wp_pred = wpl.Predict(mc.wp, x, y, N, W, NE, NW, NN, &wp_err);
      uint64_t decisions[R];
      if constexpr (R == 1) {
        // this lane's node: its property out of the ones the subtree uses
        int32_t v = 0;
#define JXL_SEL(k, expr) if (used & (1u << (k))) v = my_prop[0] == (k) ? (int32_t)(expr) : v;
        JXL_SEL(2, y) JXL_SEL(3, x)
        JXL_SEL(4, N < 0 ? 0u - (uint32_t)N : (uint32_t)N) JXL_SEL(5, W < 0 ? 0u - (uint32_t)W : (uint32_t)W) JXL_SEL(6, N) JXL_SEL(7, W)
        JXL_SEL(8, (uint32_t)W - (uint32_t)prev9) JXL_SEL(9, (uint32_t)W + (uint32_t)N - (uint32_t)NW) JXL_SEL(10, (uint32_t)W - (uint32_t)NW)
        JXL_SEL(11, (uint32_t)NW - (uint32_t)N) JXL_SEL(12, (uint32_t)N - (uint32_t)NE) JXL_SEL(13, (uint32_t)N - (uint32_t)NN)
        JXL_SEL(14, (uint32_t)W - (uint32_t)WW) JXL_SEL(15, wp_err)
#undef JXL_SEL
        decisions[0] = __ballot(v > my_val[0]);
      } else {
        if (lane == 0) {
          const int32_t grad = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
          StS<int4>(pv, make_int4(chan, (int32_t)mc.stream_id, y, x));
          StS<int4>(pv + 16, make_int4(N < 0 ? (int32_t)(0u - (uint32_t)N) : N, W < 0 ? (int32_t)(0u - (uint32_t)W) : W, N, W));
          StS<int4>(pv + 32, make_int4((int32_t)((uint32_t)W - (uint32_t)prev9), grad, (int32_t)((uint32_t)W - (uint32_t)NW), (int32_t)((uint32_t)NW - (uint32_t)N)));
          StS<int4>(pv + 48, make_int4((int32_t)((uint32_t)N - (uint32_t)NE), (int32_t)((uint32_t)N - (uint32_t)NN), (int32_t)((uint32_t)W - (uint32_t)WW), wp_err));
        }
        int32_t v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = LdS<int32_t>(pv + 4 * (uint32_t)my_prop[r]);
#pragma unroll
        for (int r = 0; r < R; r++) decisions[r] = __ballot(v[r] > my_val[r]);
      }
      uint32_t code = root_code;
      while (!(code & 0x8000)) {
        const int l = (int)(code & 63);
        const uint32_t row = code >> 6;
        uint32_t pair = 0;
        uint64_t d = 0;
#pragma unroll
        for (int r = 0; r < R; r++) if (row == (uint32_t)r) { pair = (uint32_t)__builtin_amdgcn_readlane((int)my_pair[r], l); d = decisions[r]; }
        code = ((d >> l) & 1) ? (pair & 0xFFFF) : (pair >> 16);
      }
      uint32_t n_a = 0, n_b = 1;
      int32_t n_val = 0;
      {
        const int l = (int)(code & 63);
        const uint32_t row = (code & 0x7FFF) >> 6;
#pragma unroll
        for (int r = 0; r < R; r++) if (row == (uint32_t)r) {
          n_a = (uint32_t)__builtin_amdgcn_readlane((int)leaf_a[r], l); n_b = (uint32_t)__builtin_amdgcn_readlane((int)leaf_b[r], l); n_val = __builtin_amdgcn_readlane(leaf_val[r], l);
        }
      }
      prev9 = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
      const uint32_t cluster = n_a >> 8;
      const int32_t guess = Predict(n_a & 0xFF, W, N, NW, NE, NN, WW, NEE, wp_pred);
      // ANS symbol + hybrid integer out of LDS (all lanes read the same addresses: broadcasts)
      const uint32_t res = state & 0xFFF;
      const uint32_t i = res >> (12 - la), pos_ = res & ((1u << (12 - la)) - 1);
      const uint64_t e = LdAliasAt(alias_off, T.code.wide_off, T.code.cut_off, (cluster << la) + i);
      const uint32_t cfg = LdS<uint32_t>(cfg_off + cluster * 4);
      const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
      const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
      const bool hit = pos_ >= cutoff;
      uint32_t tok = hit ? right : i;
      state = (hit ? freq1 : freq0) * (state >> 12) + (hit ? offs1 + pos_ : pos_);
      if (state < (1u << 16)) state = (state << 16) | bw.Read(16);
      const uint32_t split_exp = cfg & 0xFF;
      if (tok >= (1u << split_exp)) {
        const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
        const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - (1u << split_exp)) >> (msb + lsb))) & 31;
        const uint32_t low = tok & ((1u << lsb) - 1);
        tok >>= lsb;
        const uint32_t bits = nbits ? bw.Read((int)nbits) : 0;
        const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
        tok = (((hi << nbits) | bits) << lsb) | low;
      }
      const int32_t val = (int32_t)((uint32_t)UnpackSigned(tok) * n_b + (uint32_t)n_val + (uint32_t)guess);
      if (lane == 0) StS<int32_t>(cur + 4 * x, val);
      if (wp_here) { if (lane == 0) wpl.UpdateStores(val, x, y); wpl.NoteSample(val); }
      left2 = left; left = val;
      up0 = up1; up1 = up2; up2 = up3; up3 = up4;
    }
    WaveSync();
    for (int i = (int)lane; i < w; i += 64) StG(p + i, LdS<int32_t>(cur + 4 * i));
    const uint32_t t = prev2; prev2 = prev; prev = cur; cur = t;
  }
  const uint64_t endpos = bw.BitPos();
  br.Init(reinterpret_cast<const uint8_t*>(br.words), endpos, (uint64_t)br.wend * 4);
  WaveSync();
}

// ---- wave-wide decode of a channel under a small MA tree (round 6) ---------------------------------------------------------------------------
// The serial fast path above runs the chain on lane 0: per sample two dependent LDS round trips (property -> cluster table, cluster -> alias slot)
// and ~50 instructions, 0.25 us.  A lone wavefront issues a dependent instruction every ~4 ns whatever the number of active lanes
// (tools/microbench/chain_ops.hip), so here ALL lanes run the chain and the lanes are spent on speculation instead:
//  * lane j keeps the constant of inner node j of the channel's subtree (static splits — channel, stream, and per row the row index — resolved):
//    one compare + ballot gives every decision of the tree; with one property under test the leaf is the number of constants below the value
//    (popcount), with two the leaf lane whose path agrees with the decisions (one AND + compare per lane, ballot, find-first) — no table read;
//  * lane k owns leaf k: as soon as the ANS state of the sample is known it reads the alias slot OF ITS LEAF'S CLUSTER (StageCode's wide layout:
//    both candidates ready-made, the signed value of a complete token precomputed) — every cluster's slot in one LDS round trip, while the scalar
//    unit works out k from the sample before; `v_readlane` with k picks the winner;
//  * everything that is the same for all lanes (ANS state, bit buffer, neighbours, prediction) stays in scalar registers; the bit stream sits in a
//    VGPR, one 32-bit word per lane (a 2048-bit window), the row above and the row being decoded in VGPRs, one sample per lane (`v_readlane`,
//    a compare + select for the store): no LDS or memory access on the chain but the alias read, rows leave as coalesced 256-byte stores.
// Semantics: DecodeChunkLds / jxl_dev.h DecodeModularChannel.  Channels: leaves (predictor zero / W / clamped gradient, offset 0, multiplier 1), splits
// on the row (static per row) and on at most two of {x, N, W, W + N - NW, W - NW, NW - N}; rows up to 256 samples when the row above is needed, any
// width otherwise; at most 63 splits per row (32 with two properties), 64 leaves.
struct WaveBits {           // uniform state; `win`: this lane's word of the current 64-word window
  const uint32_t* words;
  uint32_t wend, wbase, widx;      // window = words [wbase, wbase + 64); widx: the next word to take, relative to wbase
  uint32_t win;                    // (no second window in flight: a load pending across the sample loop puts an s_waitcnt vmcnt(0) — which also waits for the row stores — into every
                                   // iteration; the switch waits for its own load instead, ~1 us per 2048 bits)
  uint64_t buf;
  int avail;
  __device__ __forceinline__ uint32_t Load(uint32_t i) const { return i < wend ? LdG(words + i) : 0u; }
  __device__ __forceinline__ void Start(const uint32_t* w, uint32_t wend_, uint64_t bit_pos, uint32_t lane) {
    words = w; wend = wend_; wbase = (uint32_t)(bit_pos >> 5); widx = 0;
    win = Load(wbase + lane);
    __builtin_amdgcn_s_waitcnt(0x0F70);
    buf = 0; avail = 0;
    Refill();
    const int skip = (int)(bit_pos & 31);
    buf >>= skip; avail -= skip;
    Refill();
  }
  __device__ __forceinline__ void Refill() {
    if (avail <= 32) {
      if (__builtin_expect(widx == 64, 0)) { const uint32_t lane = threadIdx.x & 63; wbase += 64; widx = 0; win = Load(wbase + lane); __builtin_amdgcn_s_waitcnt(0x0F70); /* vmcnt(0) here, so that none is needed in the loop */ }
      const uint32_t w = (uint32_t)__builtin_amdgcn_readlane((int)win, (int)widx);
      buf |= (uint64_t)w << avail;
      avail += 32;
      widx++;
    }
  }
  __device__ __forceinline__ uint32_t Read(int n) {      // n <= 32; at least 33 bits are buffered before and after
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; avail -= n;
    Refill();
    return v;
  }
  __device__ __forceinline__ uint64_t BitPos() const { return (uint64_t)(wbase + widx) * 32 - (uint64_t)avail; }
};
struct WaveChan {           // per row class (uniform unless noted)
  int32_t thr;              // per lane: split constant of this lane's inner node (INT_MAX beyond the last)
  bool psel;                // per lane: that node tests the second property
  uint32_t lmask, lwant;    // per lane (two properties): the inner nodes on the path to this lane's leaf / the decisions taken there
  uint32_t abase, cbase;    // per lane: LDS byte offsets of the wide table / cutoff table of this lane's leaf's cluster
  uint32_t cluster;         // per lane: that cluster
  bool is1, is5;            // per lane: this lane's leaf predicts with W / with the clamped gradient (UPRED -2)
  int32_t am0, am1;         // property value = (W & am) + this segment's vector at x
  uint32_t la, cfg_off, cfg_uniform;
};
#define SB() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ int32_t WaveShr1(int32_t v, int32_t lane0) {     // lane i <- lane i - 1 of v; lane 0 <- lane0 (DPP wave_shr:1)
  return __builtin_amdgcn_update_dpp(lane0, v, 0x138, 0xF, 0xF, false);
}
// MODE 0: one leaf; 1: one property (leaf = number of constants below the value); 2: two properties (leaf = the lane whose path matches).  UPRED: the
// predictor of every leaf (0 zero, 1 W, 5 clamped gradient) or -2: per leaf.  NEEDN: the row above exists (else N = NW = W).  vec0 / vec1: the part of
// the property values that does not depend on W, for the 64 samples of this segment; dvec: N - NW likewise; V0GRAD: vec0 is dvec and am0 all ones (the
// property is W + N - NW: the gradient predictor reuses it).
// The source is written in the order the chain should issue and pinned there (scheduling barriers, empty asm): the alias reads go out first; the
// context of the sample (leaf, prediction) and the store of the sample BEFORE are computed in their shadow; the bit buffer is topped up only after
// bits were taken.  tools/microbench/wave_ans.hip times variants of this loop: 148 -> 126 ns per sample against the plain formulation (a lone
// wavefront issues one instruction per ~2.7 ns, a dependent one per ~4.1 ns: ~38 instructions).
template <int MODE, bool NEEDN, int UPRED, bool V0GRAD>
__device__ __forceinline__ void WaveSegment(WaveBits& bits, uint32_t& state, int32_t& left, const int32_t prevv, const int32_t dvec, const int32_t vec0, const int32_t vec1, int32_t& curv, const int n, const WaveChan& wc) {
  const uint32_t la = wc.la, pmask = (1u << (12 - la)) - 1, lane = threadIdx.x & 63;
  const uint32_t sh = 12 - la;
  int32_t cur = curv;
  int32_t val_prev = 0; int xl_prev = -1;
  for (int xl = 0; xl < n; xl++) {
    // [A] alias reads
    const uint32_t slot = (state & 0xFFF) >> sh;
    const uint2 e = LdS<uint2>(wc.abase + slot * 8);
    const uint32_t cr = LdS<uint16_t>(wc.cbase + slot * 2);
    SB();
    const uint32_t pos = state & pmask, hi = state >> 12, hp = hi + pos;
    SB();
    // [C] the sample before goes to its lane
    cur = (int)lane == xl_prev ? val_prev : cur;
    asm volatile("" : "+v"(cur));
    SB();
    // [B] context
    const int32_t W = left;
    int k = 0;
    int32_t v0 = 0;
    if (MODE >= 1) v0 = V0GRAD ? (int32_t)((uint32_t)W + (uint32_t)__builtin_amdgcn_readlane(vec0, xl)) : (int32_t)((uint32_t)(W & wc.am0) + (uint32_t)__builtin_amdgcn_readlane(vec0, xl));
    if (MODE == 1) k = __builtin_popcountll(__ballot(v0 > wc.thr));
    if (MODE == 2) {
      const int32_t v1 = (int32_t)((uint32_t)(W & wc.am1) + (uint32_t)__builtin_amdgcn_readlane(vec1, xl));
      const uint32_t d = (uint32_t)__ballot((wc.psel ? v1 : v0) > wc.thr);
      k = __builtin_ctzll(__ballot((d & wc.lmask) == wc.lwant) | (1ull << 63));
    }
    int32_t grad = W;
    if (NEEDN && (UPRED == 5 || UPRED == -2)) {      // clamped gradient = median(N, W, W + N - NW)
      const int32_t N = __builtin_amdgcn_readlane(prevv, xl);
      const int32_t g0 = V0GRAD ? v0 : (int32_t)((uint32_t)W + (uint32_t)__builtin_amdgcn_readlane(dvec, xl));
      const int32_t m = min(N, W), M = max(N, W);
      grad = max(m, min(M, g0));
    }
    int32_t guess;
    if (UPRED == 0) guess = 0;
    else if (UPRED == 1) guess = W;
    else if (UPRED == 5) guess = grad;
    else guess = __builtin_amdgcn_readlane(wc.is1 ? W : (wc.is5 ? grad : 0), k);
    if (UPRED == -2) asm volatile("" : "+s"(guess)); else if (UPRED != 0) asm volatile("" : "+v"(guess));
    if (MODE >= 1) asm volatile("" : "+s"(k));
    SB();
    const bool hit = pos >= (cr & 0xFFu);
    const uint32_t cand = hit ? e.y : e.x;
    const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)cand, k);
    state = (sw & 0xFFFu) * hi + hp + ((sw >> 12) & 0xFFFu);
    int32_t v = (int32_t)sw >> 24;
    SB();
    if (state < (1u << 16)) { asm volatile("" ::: "memory"); /* (keeps this a branch: as selects it costs 13 scalar instructions on every sample) */ state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; bits.Refill(); }
    if (__builtin_expect(v == kWideEscape, 0)) {
      // the token carries extra bits (or is too large for the table's byte): the symbol again from this lane's {cutoff, aliased symbol}, then dec_ans.h's hybrid integer
      const uint32_t crk = (uint32_t)__builtin_amdgcn_readlane((int)cr, k);
      uint32_t tok = pos >= (crk & 0xFFu) ? (crk >> 8) : slot;
      uint32_t cfg = wc.cfg_uniform;
      if (cfg == 0xFFFFFFFFu) cfg = Uniform(LdS<uint32_t>(wc.cfg_off + 4 * (uint32_t)__builtin_amdgcn_readlane((int)wc.cluster, k)));
      const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
      if (tok >= split) {
        const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
        const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
        const uint32_t low = tok & ((1u << lsb) - 1);
        tok >>= lsb;
        if ((int)nbits > bits.avail) bits.Refill();
        const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
        bits.buf >>= nbits; bits.avail -= (int)nbits;
        const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
        tok = (((hb << nbits) | xb) << lsb) | low;
        bits.Refill();
      }
      v = UnpackSigned(tok);
    }
    const int32_t val = (int32_t)((uint32_t)v + (uint32_t)guess);
    left = val; val_prev = val; xl_prev = xl;
  }
  cur = (int)lane == xl_prev ? val_prev : cur;
  curv = cur;
}
#undef SB
// The channel's subtree as the wave-wide decoder wants it.  Lane 0 walks it depth first from `subroot` with the static properties resolved — channel (0), stream (1) and,
// unless `explore`, the row (2) — and writes into the wavefront's LUT region: per inner node its constant and which of the (at most two) dynamic properties it tests, per
// leaf the nodes on its path, the decisions taken there and the leaf's {predictor, cluster} word; then the header.  explore: both sides of every row split are visited and
// nothing is recorded but the header — an upper bound for every row, used once per channel to decide whether the wave-wide decoder takes it at all.
constexpr uint32_t kWaThr = 0, kWaSorted = 256, kWaPsel = 512, kWaMask = 768, kWaWant = 1024, kWaLeaf = 1280, kWaHdr = 1536;   // byte offsets in the LUT region; header: ok, ni, nl, nprops, prop0, prop1, has_y, upred, need_n
__device__ __forceinline__ bool WaveDynProp(int p) { return p == 3 || p == 6 || p == 7 || p == 9 || p == 10 || p == 11; }
__device__ void WaveAnalyse(const ModTables& T, uint32_t subroot, int chan, int32_t stream_id, int y, bool explore) {
  const uint32_t wb = T.wb, L = wb + kLutOff, stack = wb + kWorkOff + 64;
  int ok = 1, has_y = 0, upred = -1, need_n = 0, props[2] = {-1, -1};
  uint32_t ni = 0, nl = 0, np = 0, visited = 0;
  int sp = 0;
  StS<uint32_t>(stack, subroot); StS<uint32_t>(stack + 4, 0u); StS<uint32_t>(stack + 8, 0u); sp = 1;
  while (sp > 0 && ok) {
    sp--;
    uint32_t pos = LdS<uint32_t>(stack + 12 * sp);
    const uint32_t mask = LdS<uint32_t>(stack + 12 * sp + 4), want = LdS<uint32_t>(stack + 12 * sp + 8);
    TreeNode n = T.Node(pos);
    bool fork_y = false;
    while (n.prop == 0 || n.prop == 1 || n.prop == 2) {
      if (++visited > 400) { ok = 0; break; }
      if (n.prop == 2) { has_y = 1; if (explore) { fork_y = true; break; } }
      const int32_t v = n.prop == 0 ? chan : (n.prop == 1 ? stream_id : y);
      pos = v > n.val ? n.a : n.b;
      n = T.Node(pos);
    }
    if (!ok || ++visited > 400) { ok = 0; break; }
    if (fork_y) {
      if (sp + 2 > 60) { ok = 0; break; }
      StS<uint32_t>(stack + 12 * sp, n.a); StS<uint32_t>(stack + 12 * sp + 4, mask); StS<uint32_t>(stack + 12 * sp + 8, want); sp++;
      StS<uint32_t>(stack + 12 * sp, n.b); StS<uint32_t>(stack + 12 * sp + 4, mask); StS<uint32_t>(stack + 12 * sp + 8, want); sp++;
      continue;
    }
    if (n.prop < 0) {
      const int pr = (int)(n.a & 0xFF);
      if ((pr != 0 && pr != 1 && pr != 5) || n.val != 0 || n.b != 1 || nl >= 64) { ok = 0; break; }
      if (pr == 5) need_n = 1;
      if (upred == -1) upred = pr; else if (upred != pr) upred = -2;
      if (!explore) { StS<uint32_t>(L + kWaMask + 4 * nl, mask); StS<uint32_t>(L + kWaWant + 4 * nl, want); StS<uint32_t>(L + kWaLeaf + 4 * nl, n.a); }
      nl++;
      continue;
    }
    if (!WaveDynProp(n.prop) || ni >= 63 || sp + 2 > 60) { ok = 0; break; }
    if (n.prop != 3 && n.prop != 7) need_n = 1;
    uint32_t sel = 0;
    if (props[0] == n.prop || props[0] < 0) { props[0] = n.prop; sel = 0; if (np < 1) np = 1; }
    else if (props[1] == n.prop || props[1] < 0) { props[1] = n.prop; sel = 1; np = 2; }
    else { ok = 0; break; }
    const uint32_t j = ni++;
    if (!explore) { StS<int32_t>(L + kWaThr + 4 * j, n.val); StS<uint32_t>(L + kWaPsel + 4 * j, sel); }
    const uint32_t bit = j < 32 ? 1u << j : 0u;
    StS<uint32_t>(stack + 12 * sp, n.b); StS<uint32_t>(stack + 12 * sp + 4, mask | bit); StS<uint32_t>(stack + 12 * sp + 8, want); sp++;
    StS<uint32_t>(stack + 12 * sp, n.a); StS<uint32_t>(stack + 12 * sp + 4, mask | bit); StS<uint32_t>(stack + 12 * sp + 8, want | bit); sp++;
  }
  if (np == 2 && ni > 32) ok = 0;
  StS<int>(L + kWaHdr + 0, ok); StS<uint32_t>(L + kWaHdr + 4, ni); StS<uint32_t>(L + kWaHdr + 8, nl); StS<uint32_t>(L + kWaHdr + 12, np);
  StS<int>(L + kWaHdr + 16, props[0]); StS<int>(L + kWaHdr + 20, props[1]); StS<int>(L + kWaHdr + 24, has_y); StS<int>(L + kWaHdr + 28, upred); StS<int>(L + kWaHdr + 32, need_n);
}
// the leaf word {predictor, cluster << 8} that property value v reaches in a one-property subtree (static properties resolved on the way)
__device__ __forceinline__ uint32_t WaveWalkLeaf(const ModTables& T, uint32_t pos, int chan, int32_t stream_id, int y, int32_t v) {
  TreeNode n = T.Node(pos);
  for (int guard = 0; n.prop >= 0 && guard < 400; guard++) {
    const int32_t pv = n.prop == 0 ? chan : (n.prop == 1 ? stream_id : (n.prop == 2 ? y : v));
    n = T.Node(pv > n.val ? n.a : n.b);
  }
  return n.a;
}
// the part of property p that does not depend on W, for the 64 samples of a segment (lane = sample), and the mask W enters with; rows: the row above exists
__device__ __forceinline__ int32_t WavePropVec(int p, bool rows, int x0, uint32_t lane, int32_t prevv, int32_t shv, int32_t dvec, int32_t* am) {
  *am = (p == 7 || p == 9 || (p == 10 && rows) || (p == 6 && !rows)) ? -1 : 0;
  if (p == 3) return x0 + (int)lane;
  if (!rows) return 0;
  return p == 6 ? prevv : p == 9 ? dvec : p == 10 ? (int32_t)(0u - (uint32_t)shv) : p == 11 ? (int32_t)(0u - (uint32_t)dvec) : 0;
}
// All 64 lanes; WaveAnalyse(explore) said yes.  need_n: some row needs the row above (the caller checked the width).
__device__ void DecodeChannelWave(BitReaderP& br, uint32_t& state_io, const ModTables& T, const ChannelDesc& ch, uint32_t subroot_in, int chan_in, int32_t stream_in, bool has_y_in) {
  const uint32_t lane = threadIdx.x & 63, wb = T.wb, L = wb + kLutOff;
  // (everything that steers control flow into scalar registers: the compiler has to see the loops below as uniform to keep the chain there)
  const int w = (int)Uniform((uint32_t)ch.w), h = (int)Uniform((uint32_t)ch.h), chan = (int)Uniform((uint32_t)chan_in);
  const int32_t stream_id = (int32_t)Uniform((uint32_t)stream_in);
  const uint32_t subroot = Uniform(subroot_in);
  const bool has_y = Uniform(has_y_in ? 1u : 0u) != 0;
  WaveChan wc;
  wc.la = Uniform(T.code.log_alpha); wc.cfg_off = Uniform(T.code.cfg_off); wc.cfg_uniform = Uniform(T.code.cfg_uniform);
  const uint32_t wide_off = Uniform(T.code.wide_off), cut_off = Uniform(T.code.cut_off);
  wc.thr = 0x7FFFFFFF; wc.cluster = 0; wc.psel = false; wc.lmask = 0; wc.lwant = 1; wc.is1 = false; wc.is5 = false; wc.am0 = 0; wc.am1 = 0; wc.abase = wide_off; wc.cbase = cut_off;
  uint32_t state = Uniform(state_io);
  const uint64_t bp = br.BitPos();
  const uint64_t bp0 = ((uint64_t)Uniform((uint32_t)(bp >> 32)) << 32) | Uniform((uint32_t)bp);
  WaveBits bits;
  bits.Start(br.words, Uniform(br.wend), bp0, lane);
  int mode = 0, upred = 0, prop0 = -1, prop1 = -1;
  int32_t p0 = 0, p1 = 0, p2 = 0, p3 = 0;      // the row above, 64 samples per register (rows of up to 256 samples)
  int32_t first = 0;                            // first sample of the row above
  for (int y = 0; y < h; y++) {
    int32_t* p = ch.data + (size_t)y * ch.stride;
    if (y == 0 || has_y) {
      // ---- this row's subtree: constants and leaves into the lanes
      WaveSync();
      if (lane == 0) WaveAnalyse(T, subroot, chan, stream_id, y, false);
      WaveSync();
      const uint32_t ni = Uniform(LdS<uint32_t>(L + kWaHdr + 4)), np = Uniform(LdS<uint32_t>(L + kWaHdr + 12));
      prop0 = (int)Uniform((uint32_t)LdS<int>(L + kWaHdr + 16)); prop1 = (int)Uniform((uint32_t)LdS<int>(L + kWaHdr + 20));
      upred = (int)Uniform((uint32_t)LdS<int>(L + kWaHdr + 28));
      mode = (int)np;
      uint32_t leaf = 0;
      wc.thr = lane < ni ? LdS<int32_t>(L + kWaThr + 4 * lane) : 0x7FFFFFFF;
      if (mode == 2) {
        const uint32_t nl = Uniform(LdS<uint32_t>(L + kWaHdr + 8));
        wc.psel = lane < ni && LdS<uint32_t>(L + kWaPsel + 4 * lane) != 0;
        wc.lmask = lane < nl ? LdS<uint32_t>(L + kWaMask + 4 * lane) : 0u;
        wc.lwant = lane < nl ? LdS<uint32_t>(L + kWaWant + 4 * lane) : 1u;
        leaf = LdS<uint32_t>(L + kWaLeaf + 4 * min(lane, nl - 1));
      } else if (mode == 1) {
        // interval k = (number of constants below the value): lane k walks the subtree once with a value of its interval
        const int32_t t = wc.thr;
        uint32_t rank = 0;
        for (uint32_t i = 0; i < ni; i++) { const int32_t ti = __builtin_amdgcn_readlane(t, (int)i); rank += (ti < t || (ti == t && i < lane)) ? 1u : 0u; }
        if (lane < ni) StS<int32_t>(L + kWaSorted + 4 * rank, t);
        WaveSync();
        const uint32_t kk = min(lane, ni);
        const int32_t rep = kk == 0 ? LdS<int32_t>(L + kWaSorted) : (int32_t)((uint32_t)LdS<int32_t>(L + kWaSorted + 4 * (kk - 1)) + 1u);
        leaf = WaveWalkLeaf(T, subroot, chan, stream_id, y, rep);
      } else leaf = LdS<uint32_t>(L + kWaLeaf);
      wc.cluster = leaf >> 8;
      wc.is1 = (leaf & 0xFF) == 1; wc.is5 = (leaf & 0xFF) == 5;
      wc.abase = wide_off + ((wc.cluster << wc.la) << 3); wc.cbase = cut_off + ((wc.cluster << wc.la) << 1);
    }
    const bool rows = y > 0;
    int32_t left = 0;
    if (y > 0) left = w <= 256 ? __builtin_amdgcn_readlane(p0, 0) : first;       // W of the first sample = N (rows wider than 256 samples only get here when no row needs N but this)
    int32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    int32_t nw_in = left;
    for (int x0 = 0; x0 < w; x0 += 64) {
      const int n = min(64, w - x0), seg = (x0 >> 6) & 3;
      const int32_t prevv = seg == 0 ? p0 : seg == 1 ? p1 : seg == 2 ? p2 : p3;
      const int32_t shv = WaveShr1(prevv, nw_in), dvec = prevv - shv;
      nw_in = __builtin_amdgcn_readlane(prevv, 63);
      int32_t am0 = 0, am1 = 0;
      const int32_t vec0 = WavePropVec(prop0, rows, x0, lane, prevv, shv, dvec, &am0), vec1 = WavePropVec(prop1, rows, x0, lane, prevv, shv, dvec, &am1);
      wc.am0 = (int32_t)Uniform((uint32_t)am0); wc.am1 = (int32_t)Uniform((uint32_t)am1);
      int32_t curv = 0;
#define JXL_WSEG(M, NN, UP, VG) WaveSegment<M, NN, UP, VG>(bits, state, left, prevv, dvec, vec0, vec1, curv, n, wc)
#define JXL_WSEG_UP(M, NN, VG) do { if (upred == 0) JXL_WSEG(M, NN, 0, VG); else if (upred == 1) JXL_WSEG(M, NN, 1, VG); else if (upred == 5) JXL_WSEG(M, NN, 5, VG); else JXL_WSEG(M, NN, -2, VG); } while (0)
      if (mode == 0) { if (rows) JXL_WSEG_UP(0, true, false); else JXL_WSEG_UP(0, false, false); }
      else if (mode == 1) {
        if (rows) { if (prop0 == 9) JXL_WSEG_UP(1, true, true); else JXL_WSEG_UP(1, true, false); }
        else JXL_WSEG_UP(1, false, false);
      } else { if (rows) JXL_WSEG_UP(2, true, false); else JXL_WSEG_UP(2, false, false); }
#undef JXL_WSEG_UP
#undef JXL_WSEG
      if ((int)lane < n) StG(p + x0 + (int)lane, curv);
      if (x0 == 0) first = __builtin_amdgcn_readlane(curv, 0);
      if (seg == 0) c0 = curv; else if (seg == 1) c1 = curv; else if (seg == 2) c2 = curv; else c3 = curv;
    }
    p0 = c0; p1 = c1; p2 = c2; p3 = c3;
  }
  state_io = state;
  const uint64_t endpos = bits.BitPos();
  br.Init(reinterpret_cast<const uint8_t*>(br.words), endpos, (uint64_t)br.wend * 4);
  WaveSync();
}

// ---- one token of a wave-wide decoder without speculation (the cluster is known): HfDecodeWaveKernel's count tokens, the big-tree form of DecodeChannelWaveGen
struct WaveTok { uint32_t u; int32_t v; };      // a decoded hybrid integer and its UnpackSigned()
// one token under a known (uniform) cluster base: the non-speculative form (the "number of non-zeros" token of a block)
__device__ __forceinline__ WaveTok WaveTokenAt(WaveBits& bits, uint32_t& state, uint32_t abase, uint32_t cbase, uint32_t cfg, uint32_t la) {
  const uint32_t pmask = (1u << (12 - la)) - 1;
  const uint32_t slot = (state & 0xFFF) >> (12 - la), pos = state & pmask, hi = state >> 12;
  const uint2 e = LdS<uint2>(abase + slot * 8);
  const uint32_t cr = Uniform(LdS<uint16_t>(cbase + slot * 2));
  const bool hit = pos >= (cr & 0xFFu);
  const uint32_t sw = Uniform(hit ? e.y : e.x);
  state = (sw & 0xFFFu) * hi + hi + pos + ((sw >> 12) & 0xFFFu);
  int32_t v = (int32_t)sw >> 24;
  if (state < (1u << 16)) { state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; bits.Refill(); }
  WaveTok t;
  if (v != kWideEscape) { t.v = v; t.u = (uint32_t)((v << 1) ^ (v >> 31)); return t; }
  uint32_t tok = hit ? (cr >> 8) : slot;
  const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
  if (tok >= split) {
    const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
    const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
    const uint32_t low = tok & ((1u << lsb) - 1);
    tok >>= lsb;
    if ((int)nbits > bits.avail) bits.Refill();
    const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
    bits.buf >>= nbits; bits.avail -= (int)nbits;
    const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
    tok = (((hb << nbits) | xb) << lsb) | low;
    bits.Refill();
  }
  t.u = tok; t.v = UnpackSigned(tok);
  return t;
}
// ---- wave-wide decode of a channel under a general MA tree, with or without the weighted predictor (round 6) ---------------------------------------------------
// Two tree shapes.  THRESH: every split tests property 15 — the largest neighbouring error of the weighted predictor — and every leaf predicts with it: the LF
// coefficients of a default-effort cjxl encode (enc_modular.cc "WP fixed DC"); the leaf is a popcount as in DecodeChannelWave.  GEN: what cjxl's lossless modes
// write — up to 63 splits on any of {x, N, W, W+N-NW, W-NW, NW-N, N-NE, N-NN, weighted-predictor error} (row / channel / stream splits resolved per row), leaves
// predicting with zero / W / clamped gradient / weighted predictor: lane j evaluates ITS node's property as a per-lane linear form of the (uniform) neighbours —
// six multiply-adds for all 64 nodes at once, whatever the mix of properties —, one compare + ballot yields every decision, the leaf lane whose path agrees is
// found with one AND + compare per lane, ballot and find-first; the prediction is the leaf lane's pick of the candidates.
// The entropy chain is DecodeChannelWave's (lanes = leaves, alias slots of all clusters in one LDS round trip).  The weighted predictor (context_predict.h
// weighted::State, jxl_dev.h WPState) is spread over the lanes of every quad: lane q of a quad owns sub-predictor q — its error sums (one LDS read per sample: the
// stored error at x + 1 of the row above; the additions of the reference's "+= error at x + 1" live in register carries), its weight (division table in LDS), its
// prediction (a per-lane linear form of the neighbours and true errors) — and the sums over the four are DPP quad permutes.  True errors and samples of the rows
// above sit in VGPRs, one per lane, like DecodeChannelWave's rows.  32-bit arithmetic (exact while |sample| <= 4095, as WPStateLds::PredictT<int32_t>; without the
// predictor while |sample| < 2^22: the 24-bit multiply-adds): the first larger sample ends the attempt and the caller decodes the channel again the general way.
// Rows up to 256 samples.
struct WaveWpLane { int32_t kW, kNE, kN, cW, cN, cNE, cNW, cNN, cNWW; uint32_t wmax; };
struct WaveGenLane {        // per lane: the linear form of this lane's inner node's property, this lane's leaf
  int32_t aW, aN, aNW, aNE, aNN, aE, aX, c0, thr;
  uint32_t mlo, mhi, wlo, whi;
  bool is1, is5, is6;
};
__device__ __forceinline__ int32_t QuadSumI(int32_t v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
  return v;
}
__device__ __forceinline__ int32_t WaveShl1(int32_t v, int32_t lane63) {    // lane i <- lane i + 1 of v; lane 63 <- lane63 (DPP wave_shl:1)
  return __builtin_amdgcn_update_dpp(lane63, v, 0x130, 0xF, 0xF, false);
}
struct WaveWpRow {          // carries of one row (uniform unless noted)
  int32_t left, Nprev, teW, teNprev;
  int32_t e0prev, aprev, e1x;          // per lane (sub-predictor lane & 3): error of the sample before, A of the sample before, stored error of the row above at x
  int32_t toobig;
};
// TREE 0: thresholds on property 15 (popcount); 1: general, up to 63 splits (linear forms, path match); 2: big — up to 192 splits per property (768 in all) of cjxl's lossless set
// {W+N-NW, W-NW, NW-N, N-NE, N-NN, weighted-predictor error}, hundreds per subtree: the lanes hold the nodes grouped by property (three register rows of 64 per property), a
// compare per row yields every decision, every node lane writes its chosen child into a next-pointer table in LDS, and the walk from the root is a chain of 16 dependent LDS
// reads (leaves point at themselves) that the compiler interleaves with the weighted predictor's arithmetic; the entropy symbol follows without speculation.
constexpr int kBigTreeNodes = 160;      // frames with larger MA trees take the Modular group kernel's big-tree instantiation
constexpr int kBigRows = 18, kBigRowsPerProp = 3, kBigPerProp = 64 * kBigRowsPerProp, kBigNodes = 768;
struct WaveBigLane { int32_t thr[kBigRows]; uint32_t a[kBigRows], b[kBigRows], addr[kBigRows]; uint32_t next_off, root, node_base, wide_off, cut_off, nrows; };
template <bool ROW0, bool LAST, bool USE_WP, int TREE>
__device__ __forceinline__ void WaveGenSample(WaveBits& bits, uint32_t& state, WaveWpRow& r, const int xl, const uint32_t x, const int32_t p1v, const int32_t p1s, const int32_t p2v, const int32_t te1v, const int32_t te1s,
                                             int32_t& curv, int32_t& tecur, const uint32_t e1, const uint32_t e0, const uint32_t div_off, const WaveWpLane& L, const WaveGenLane& G, const WaveBigLane& BG, const WaveChan& wc) {
  constexpr bool GEN = TREE == 1;
  const uint32_t la = wc.la, pmask = (1u << (12 - la)) - 1, lane = threadIdx.x & 63;
  // --- alias reads (the speculative forms: every leaf's cluster at once)
  const uint32_t slot = (state & 0xFFF) >> (12 - la);
  uint2 e = make_uint2(0, 0);
  uint32_t cr = 0;
  if (TREE != 2) { e = LdS<uint2>(wc.abase + slot * 8); cr = LdS<uint16_t>(wc.cbase + slot * 2); }
  const uint32_t pos = state & pmask, hi = state >> 12, hp = hi + pos;
  // --- neighbours and true errors
  const int32_t W = r.left;
  int32_t N = W, NE = W, NW = W, NN = W, teN = 0, teNE = 0, teNW = 0;
  if (!ROW0) {
    N = __builtin_amdgcn_readlane(p1v, xl); NE = LAST ? N : __builtin_amdgcn_readlane(p1s, xl); NW = r.Nprev; NN = __builtin_amdgcn_readlane(p2v, xl);
    if (USE_WP) { teN = __builtin_amdgcn_readlane(te1v, xl); teNE = LAST ? teN : __builtin_amdgcn_readlane(te1s, xl); teNW = r.teNprev; }
  }
  int32_t pred = 0, perr = 0, predi = 0;
  if (USE_WP) {
    const int32_t teW = r.teW;
    const int32_t W8 = W * 8, N8 = N * 8, NE8 = NE * 8, NW8 = NW * 8, NN8 = NN * 8;
    // --- error sums of this lane's sub-predictor: A (at N, includes the error of W), B (at NW = the A of the sample before), C (at NE)
    const int32_t A = r.e1x + r.e0prev;
    const int32_t C = LAST ? A : LdS<int32_t>(e1 + 4 * (x + 1));
    const uint32_t sum = (uint32_t)A + (uint32_t)r.aprev + (uint32_t)C;
    int shift = 26 - __clz((int)(sum + 1));          // floor(log2(sum + 1)) - 5
    if (shift < 0) shift = 0;
    const uint32_t quot = LdS<uint32_t>(div_off + 4 * (sum >> shift));
    uint32_t wgt = 4 + ((L.wmax * quot) >> shift);
    const uint32_t wsum = (uint32_t)QuadSumI((int32_t)wgt);
    wgt >>= (27 - __clz((int)wsum));                 // floor(log2(wsum)) - 4
    const uint32_t wsum2 = (uint32_t)QuadSumI((int32_t)wgt);
    const uint32_t inv = LdS<uint32_t>(div_off + 4 * (wsum2 - 1));
    // --- this lane's sub-prediction
    const int32_t lin = L.cW * teW + L.cN * teN + L.cNE * teNE + L.cNW * teNW + L.cNN * (NN8 - N8) + L.cNWW * (NW8 - W8);
    predi = L.kW * W8 + L.kNE * NE8 + L.kN * N8 - (lin >> 5);
    const int32_t sump = QuadSumI(predi * (int32_t)wgt) + (int32_t)(wsum2 >> 1) - 1;
    pred = (int32_t)(((int64_t)sump * (int64_t)inv) >> 24);
    if (!(((teN ^ teW) | (teN ^ teNW)) > 0)) {
      const int32_t mx = max(W8, max(NE8, N8)), mn = min(W8, min(NE8, N8));
      pred = max(mn, min(mx, pred));
    }
    // --- property 15: the true error of largest magnitude among W, N, NW, NE (the first of equals)
    perr = teW;
    if (abs(teN) > abs(perr)) perr = teN;
    if (abs(teNW) > abs(perr)) perr = teNW;
    if (abs(teNE) > abs(perr)) perr = teNE;
    r.aprev = A; r.e1x = C;
  }
  int k = 0;
  int32_t guess;
  const int32_t wpguess = (pred + 3) >> 3;
  int32_t v = 0;
  if (TREE == 2) {
    // every node's decision, its chosen child into the next-pointer table, the walk
    const int32_t pv[6] = {(int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW), (int32_t)((uint32_t)W - (uint32_t)NW), (int32_t)((uint32_t)NW - (uint32_t)N),
                           (int32_t)((uint32_t)N - (uint32_t)NE), (int32_t)((uint32_t)N - (uint32_t)NN), perr};
#pragma unroll
    for (int rr = 0; rr < kBigRows; rr++) StS<uint32_t>(BG.addr[rr], pv[rr / kBigRowsPerProp] > BG.thr[rr] ? BG.a[rr] : BG.b[rr]);      // (unused node lanes write the spare entry)
    // (the table holds ADDRESSES of entries — a hop is one LDS read with nothing to compute in between)
    uint32_t at = BG.next_off + BG.root * 4;
#pragma unroll
    for (int i = 0; i < 12; i++) at = LdS<uint32_t>(at);
    uint4 nd = LdS<uint4>(BG.node_base + (at - BG.next_off) * 4);
    while ((int32_t)Uniform(nd.x) >= 0) { at = LdS<uint32_t>(at); nd = LdS<uint4>(BG.node_base + (at - BG.next_off) * 4); }      // (paths longer than 12)
    const uint32_t leaf = Uniform(nd.z), pk = leaf & 0xFF, cl = leaf >> 8;
    const int32_t m = min(N, W), M = max(N, W);
    const int32_t grad = max(m, min(M, pv[0]));
    guess = pk == 6 ? wpguess : (pk == 5 ? grad : (pk == 1 ? W : 0));
    const uint32_t cfg = wc.cfg_uniform != 0xFFFFFFFFu ? wc.cfg_uniform : Uniform(LdS<uint32_t>(wc.cfg_off + 4 * cl));
    v = WaveTokenAt(bits, state, BG.wide_off + ((cl << la) << 3), BG.cut_off + ((cl << la) << 1), cfg, la).v;
  } else if (!GEN) {
    k = __builtin_popcountll(__ballot(perr > wc.thr));
    guess = wpguess;
  } else {
    // every inner node's property value in its lane (24-bit multiply-adds: the samples are checked below), all decisions, the leaf whose path agrees
    int32_t pv = G.c0 + __mul24(G.aX, (int32_t)x);
    pv += __mul24(G.aN, N); pv += __mul24(G.aNW, NW); pv += __mul24(G.aNE, NE); pv += __mul24(G.aNN, NN);
    if (USE_WP) pv += __mul24(G.aE, perr);
    pv += __mul24(G.aW, W);
    const uint64_t d = __ballot(pv > G.thr);
    const uint32_t dlo = (uint32_t)d, dhi = (uint32_t)(d >> 32);
    k = __builtin_ctzll(__ballot((dlo & G.mlo) == G.wlo && (dhi & G.mhi) == G.whi) | (1ull << 63));
    const int32_t m = min(N, W), M = max(N, W);
    const int32_t grad = max(m, min(M, (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW)));
    guess = __builtin_amdgcn_readlane(G.is6 ? wpguess : (G.is5 ? grad : (G.is1 ? W : 0)), k);
  }
  // --- ANS symbol
  if (TREE != 2) {
  const bool hit = pos >= (cr & 0xFFu);
  const uint32_t cand = hit ? e.y : e.x;
  const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)cand, k);
  state = (sw & 0xFFFu) * hi + hp + ((sw >> 12) & 0xFFFu);
  v = (int32_t)sw >> 24;
  if (state < (1u << 16)) { asm volatile("" ::: "memory"); state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; bits.Refill(); }
  if (__builtin_expect(v == kWideEscape, 0)) {
    const uint32_t crk = (uint32_t)__builtin_amdgcn_readlane((int)cr, k);
    uint32_t tok = pos >= (crk & 0xFFu) ? (crk >> 8) : slot;
    uint32_t cfg = wc.cfg_uniform;
    if (cfg == 0xFFFFFFFFu) cfg = Uniform(LdS<uint32_t>(wc.cfg_off + 4 * (uint32_t)__builtin_amdgcn_readlane((int)wc.cluster, k)));
    const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
    if (tok >= split) {
      const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
      const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
      const uint32_t low = tok & ((1u << lsb) - 1);
      tok >>= lsb;
      if ((int)nbits > bits.avail) bits.Refill();
      const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
      bits.buf >>= nbits; bits.avail -= (int)nbits;
      const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
      tok = (((hb << nbits) | xb) << lsb) | low;
      bits.Refill();
    }
    v = UnpackSigned(tok);
  }
  }
  const int32_t val = (int32_t)((uint32_t)v + (uint32_t)guess);
  if (USE_WP) { if ((uint32_t)(val + 4095) > 8190u) r.toobig = 1; }
  else if ((uint32_t)(val + (1 << 22)) > (1u << 23)) r.toobig = 1;
  curv = (int)lane == xl ? val : curv;
  if (USE_WP) {
    // --- the sample's errors: magnitudes per sub-predictor (stored for the row below, carried for the sample to the right), true error
    const int32_t v8 = val * 8;
    const int32_t err = (abs(predi - v8) + 3) >> 3;
    StS<int32_t>(e0 + 4 * x, err);
    const int32_t te = pred - v8;
    tecur = (int)lane == xl ? te : tecur;
    r.teW = te; r.teNprev = teN; r.e0prev = err;
  }
  r.left = val; r.Nprev = N;
}
// The general shape's analysis (lane 0): as WaveAnalyse, for the wider property / predictor sets and 64-bit paths.  Per inner node j: property and constant; per leaf:
// path mask / decisions (two words each) and the {predictor, cluster} word.  Header: ok, ni, nl, has_y, uses_wp, thresh (every split on property 15, every leaf the
// weighted predictor: the popcount form).
constexpr uint32_t kWgProp = kWaPsel, kWgMaskHi = kWaSorted, kWgWantHi = 1600;
__device__ __forceinline__ bool WaveGenProp(int p) { return p == 3 || p == 6 || p == 7 || (p >= 9 && p <= 13) || p == 15; }
__device__ void WaveAnalyseGen(const ModTables& T, uint32_t subroot, int chan, int32_t stream_id, int y, bool explore) {
  const uint32_t wb = T.wb, L = wb + kLutOff, stack = wb + kWorkOff + 64;    // stack entries: node, mask lo, mask hi, want lo, want hi
  int ok = 1, has_y = 0, uses_wp = 0, thresh = 1;
  uint32_t ni = 0, nl = 0, visited = 0;
  int sp = 0;
  StS<uint32_t>(stack, subroot); StS<uint32_t>(stack + 4, 0u); StS<uint32_t>(stack + 8, 0u); StS<uint32_t>(stack + 12, 0u); StS<uint32_t>(stack + 16, 0u); sp = 1;
  while (sp > 0 && ok) {
    sp--;
    uint32_t pos = LdS<uint32_t>(stack + 20 * sp);
    const uint32_t mlo = LdS<uint32_t>(stack + 20 * sp + 4), mhi = LdS<uint32_t>(stack + 20 * sp + 8), wlo = LdS<uint32_t>(stack + 20 * sp + 12), whi = LdS<uint32_t>(stack + 20 * sp + 16);
    TreeNode n = T.Node(pos);
    bool fork_y = false;
    while (n.prop == 0 || n.prop == 1 || n.prop == 2) {
      if (++visited > 600) { ok = 0; break; }
      if (n.prop == 2) { has_y = 1; if (explore) { fork_y = true; break; } }
      const int32_t v = n.prop == 0 ? chan : (n.prop == 1 ? stream_id : y);
      pos = v > n.val ? n.a : n.b;
      n = T.Node(pos);
    }
    if (!ok || ++visited > 600) { ok = 0; break; }
    auto push = [&](uint32_t node, uint32_t a0, uint32_t a1, uint32_t b0, uint32_t b1) {
      StS<uint32_t>(stack + 20 * sp, node); StS<uint32_t>(stack + 20 * sp + 4, a0); StS<uint32_t>(stack + 20 * sp + 8, a1); StS<uint32_t>(stack + 20 * sp + 12, b0); StS<uint32_t>(stack + 20 * sp + 16, b1); sp++;
    };
    if (fork_y) {
      if (sp + 2 > 38) { ok = 0; break; }
      push(n.a, mlo, mhi, wlo, whi); push(n.b, mlo, mhi, wlo, whi);
      continue;
    }
    if (n.prop < 0) {
      const int pr = (int)(n.a & 0xFF);
      if ((pr != 0 && pr != 1 && pr != 5 && pr != 6) || n.val != 0 || n.b != 1 || nl >= 64) { ok = 0; break; }
      if (pr == 6) uses_wp = 1; else thresh = 0;
      if (!explore) { StS<uint32_t>(L + kWaMask + 4 * nl, mlo); StS<uint32_t>(L + kWgMaskHi + 4 * nl, mhi); StS<uint32_t>(L + kWaWant + 4 * nl, wlo); StS<uint32_t>(L + kWgWantHi + 4 * nl, whi); StS<uint32_t>(L + kWaLeaf + 4 * nl, n.a); }
      nl++;
      continue;
    }
    if (!WaveGenProp(n.prop) || ni >= 63 || sp + 2 > 38) { ok = 0; break; }
    if (n.prop == 15) uses_wp = 1; else thresh = 0;
    const uint32_t j = ni++;
    if (!explore) { StS<int32_t>(L + kWaThr + 4 * j, n.val); StS<uint32_t>(L + kWgProp + 4 * j, (uint32_t)n.prop); }
    const uint32_t blo = j < 32 ? 1u << j : 0u, bhi = j >= 32 ? 1u << (j - 32) : 0u;
    push(n.b, mlo | blo, mhi | bhi, wlo, whi);
    push(n.a, mlo | blo, mhi | bhi, wlo | blo, whi | bhi);
  }
  StS<int>(L + kWaHdr + 0, ok); StS<uint32_t>(L + kWaHdr + 4, ni); StS<uint32_t>(L + kWaHdr + 8, nl); StS<int>(L + kWaHdr + 24, has_y); StS<int>(L + kWaHdr + 36, uses_wp); StS<int>(L + kWaHdr + 40, thresh);
}
// The big shape's analysis (lane 0): the subtree's dynamic nodes grouped by property — node indices, up to 128 per property, at the start of the LUT region —, the
// next-pointer entries that never change (leaves point at themselves, static splits at the child their property picks).  Header: ok, has_y, uses_wp, counts per property.
constexpr uint32_t kWbList = 0, kWbCnt = kWaHdr + 48, kWbOff = kWaHdr + 72, kWbNextEntries = 1025;       // (entry 1024: where the unused node lanes write)
static_assert(kWbList + 2 * kBigNodes <= kWaHdr, "node list in front of the header");
__device__ __forceinline__ int WaveBigProp(int p) { return p >= 9 && p <= 13 ? p - 9 : (p == 15 ? 5 : -1); }
__device__ void WaveAnalyseBig(const ModTables& T, uint32_t next_off, int chan, int32_t stream_id, int y, bool explore) {
  const uint32_t wb = T.wb, L = wb + kLutOff, stack = wb + kWorkOff + 64;
  int ok = 1, has_y = 0, uses_wp = 0;
  uint32_t cnt[6] = {0, 0, 0, 0, 0, 0}, off[6] = {0, 0, 0, 0, 0, 0};
  // two walks: the first counts the nodes per property (and fills the constant next-pointer entries), the second puts the node indices into their property's stretch of the list
  for (int pass = 0; pass < (explore ? 1 : 2) && ok; pass++) {
    uint32_t visited = 0, placed[6] = {0, 0, 0, 0, 0, 0};
    int sp = 0;
    StS<uint32_t>(stack, 0u); sp = 1;
    while (sp > 0 && ok) {
      const uint32_t pos = LdS<uint32_t>(stack + 4 * --sp);
      if (pos >= 1024 || ++visited > 2048) { ok = 0; break; }
      const TreeNode n = T.Node(pos);
      if (n.prop < 0) {
        const int pr = (int)(n.a & 0xFF);
        if ((pr != 0 && pr != 1 && pr != 5 && pr != 6) || n.val != 0 || n.b != 1) { ok = 0; break; }
        if (pr == 6) uses_wp = 1;
        if (!explore && pass == 0) StS<uint32_t>(next_off + 4 * pos, next_off + 4 * pos);
        continue;
      }
      if (sp + 2 > 190) { ok = 0; break; }
      if (n.prop == 0 || n.prop == 1 || n.prop == 2) {
        if (n.prop == 2) has_y = 1;
        if (n.prop == 2 && explore) { StS<uint32_t>(stack + 4 * sp++, n.a); StS<uint32_t>(stack + 4 * sp++, n.b); continue; }
        const int32_t v = n.prop == 0 ? chan : (n.prop == 1 ? stream_id : y);
        const uint32_t child = v > n.val ? n.a : n.b;
        if (!explore && pass == 0) StS<uint32_t>(next_off + 4 * pos, next_off + 4 * child);
        StS<uint32_t>(stack + 4 * sp++, child);
        continue;
      }
      const int pi = WaveBigProp(n.prop);
      if (pi < 0) { ok = 0; break; }
      if (pi == 5) uses_wp = 1;
      if (pass == 0) { if (++cnt[pi] > (uint32_t)kBigPerProp) { ok = 0; break; } }
      else StS<uint16_t>(L + kWbList + 2 * (off[pi] + placed[pi]++), (uint16_t)pos);
      StS<uint32_t>(stack + 4 * sp++, n.a); StS<uint32_t>(stack + 4 * sp++, n.b);
    }
    if (pass == 0) {
      uint32_t total = 0;
      for (int i = 0; i < 6; i++) { off[i] = total; total += cnt[i]; }
      if (total > (uint32_t)kBigNodes) ok = 0;
    }
  }
  StS<int>(L + kWaHdr + 0, ok); StS<int>(L + kWaHdr + 24, has_y); StS<int>(L + kWaHdr + 36, uses_wp);
  for (int i = 0; i < 6; i++) { StS<uint32_t>(L + kWbCnt + 4 * i, cnt[i]); StS<uint32_t>(L + kWbOff + 4 * i, off[i]); }
}
// All 64 lanes; WaveAnalyseGen(explore) said yes.  false: a sample beyond the range of the 32-bit arithmetic (nothing of `br` / `state_io` was touched: the caller
// decodes the channel again the general way).
// BIG: the big-tree form (WaveAnalyseBig said yes; the next-pointer table takes the wavefront's bit-stream window / row buffers, which this decoder does not use)
template <bool USE_WP, bool BIG>
__device__ bool DecodeChannelWaveGen(BitReaderP& br, uint32_t& state_io, const ModTables& T, const ChannelDesc& ch, const WPHeader& hdr, int chan_in, int32_t stream_in, bool has_y_in) {
  const uint32_t lane = threadIdx.x & 63, wb = T.wb, LR = wb + kLutOff;
  static_assert(kWinOff + kWbNextEntries * 4 <= kWaveLds, "next-pointer table in the per-wavefront LDS region");
  WaveBigLane BG;
  BG.next_off = wb + kWinOff; BG.root = 0; BG.node_base = Uniform(T.node_base); BG.wide_off = Uniform(T.code.wide_off); BG.cut_off = Uniform(T.code.cut_off); BG.nrows = 0;
#pragma unroll
  for (int rr = 0; rr < kBigRows; rr++) { BG.thr[rr] = 0x7FFFFFFF; BG.a[rr] = 0; BG.b[rr] = 0; BG.addr[rr] = BG.next_off + 1024 * 4; }
  const int w = (int)Uniform((uint32_t)ch.w), h = (int)Uniform((uint32_t)ch.h), chan = (int)Uniform((uint32_t)chan_in);
  const int32_t stream_id = (int32_t)Uniform((uint32_t)stream_in);
  const bool has_y = Uniform(has_y_in ? 1u : 0u) != 0;
  WaveChan wc;
  wc.la = Uniform(T.code.log_alpha); wc.cfg_off = Uniform(T.code.cfg_off); wc.cfg_uniform = Uniform(T.code.cfg_uniform);
  const uint32_t wide_off = Uniform(T.code.wide_off), cut_off = Uniform(T.code.cut_off);
  wc.thr = 0x7FFFFFFF; wc.cluster = 0; wc.abase = wide_off; wc.cbase = cut_off;
  WaveGenLane G;
  G.aW = G.aN = G.aNW = G.aNE = G.aNN = G.aE = G.aX = G.c0 = 0; G.thr = 0x7FFFFFFF; G.mlo = G.mhi = 0; G.wlo = 1; G.whi = 0; G.is1 = G.is5 = G.is6 = false;
  // weighted-predictor state: WPStateLds' layout (error rows of sub-predictor i at array 1 + i, two rows of w + 2 ints each), zeroed; the division table behind it
  const uint32_t wp_base = USE_WP ? Uniform(T.wp_off) : 0u, div_off = wp_base + kWpLdsBytes - 256;
  const uint32_t q = lane & 3, rowb = (uint32_t)(w + 2) * 4;
  const uint32_t erows = wp_base + (1 + q) * rowb * 2;
  WaveWpLane L;
  L.kW = L.kNE = L.kN = L.cW = L.cN = L.cNE = L.cNW = L.cNN = L.cNWW = 0; L.wmax = 0;
  if (USE_WP) {
    for (uint32_t i = lane; i < WPStateLds::Bytes(w) / 4; i += 64) StS<int32_t>(wp_base + i * 4, 0);
    StS<uint32_t>(div_off + lane * 4, (1u << 24) / (lane + 1));
    const int32_t p1 = hdr.p1, p2 = hdr.p2;
    L.kW = q == 0 || q == 2 ? 1 : 0; L.kNE = q == 0 ? 1 : 0; L.kN = q == 0 ? -1 : (q == 2 ? 0 : 1);
    L.cW = q == 1 ? p1 : (q == 2 ? p2 : 0);
    L.cN = q == 1 ? p1 : (q == 2 ? p2 : (q == 3 ? hdr.p3[1] : 0));
    L.cNE = q == 1 ? p1 : (q == 3 ? hdr.p3[2] : 0);
    L.cNW = q == 2 ? p2 : (q == 3 ? hdr.p3[0] : 0);
    L.cNN = q == 3 ? hdr.p3[3] : 0;
    L.cNWW = q == 3 ? hdr.p3[4] : 0;
    L.wmax = (uint32_t)hdr.w[q];
  }
  uint32_t state = Uniform(state_io);
  const uint64_t bp = br.BitPos();
  const uint64_t bp0 = ((uint64_t)Uniform((uint32_t)(bp >> 32)) << 32) | Uniform((uint32_t)bp);
  WaveBits bits;
  bits.Start(br.words, Uniform(br.wend), bp0, lane);
  int32_t p1r[4] = {0, 0, 0, 0}, p2r[4] = {0, 0, 0, 0}, t1r[4] = {0, 0, 0, 0};     // samples of the row above / two above, true errors of the row above
  WaveWpRow r;
  r.toobig = 0; r.teW = 0; r.teNprev = 0; r.e0prev = 0; r.aprev = 0; r.e1x = 0;
  const int nseg = (w + 63) >> 6;
  bool thresh = false;
  for (int y = 0; y < h && !r.toobig; y++) {
    int32_t* p = ch.data + (size_t)y * ch.stride;
    if (BIG && (y == 0 || has_y)) {
      // ---- this row's subtree: nodes grouped by property into the lanes, the constant entries of the next-pointer table
      WaveSync();
      if (lane == 0) WaveAnalyseBig(T, BG.next_off, chan, stream_id, y, false);
      WaveSync();
#pragma unroll
      for (int rr = 0; rr < kBigRows; rr++) {
        const uint32_t pi = (uint32_t)rr / kBigRowsPerProp, slot = (uint32_t)(rr % kBigRowsPerProp) * 64 + lane;
        const uint32_t cnt = Uniform(LdS<uint32_t>(LR + kWbCnt + 4 * pi)), first = Uniform(LdS<uint32_t>(LR + kWbOff + 4 * pi));
        BG.thr[rr] = 0x7FFFFFFF; BG.a[rr] = 0; BG.b[rr] = 0; BG.addr[rr] = BG.next_off + 1024 * 4;
        if (slot < cnt) {
          const uint32_t j = LdS<uint16_t>(LR + kWbList + 2 * (first + slot));
          const TreeNode n = T.Node(j);
          BG.thr[rr] = n.val; BG.a[rr] = BG.next_off + n.a * 4; BG.b[rr] = BG.next_off + n.b * 4; BG.addr[rr] = BG.next_off + j * 4;
        }
      }
      WaveSync();
    }
    if (!BIG && (y == 0 || has_y)) {
      // ---- this row's subtree: linear forms, constants and leaves into the lanes
      WaveSync();
      if (lane == 0) WaveAnalyseGen(T, 0, chan, stream_id, y, false);
      WaveSync();
      const uint32_t ni = Uniform(LdS<uint32_t>(LR + kWaHdr + 4)), nl = Uniform(LdS<uint32_t>(LR + kWaHdr + 8));
      thresh = USE_WP && Uniform((uint32_t)LdS<int>(LR + kWaHdr + 40)) != 0;
      const int32_t t = lane < ni ? LdS<int32_t>(LR + kWaThr + 4 * lane) : 0x7FFFFFFF;
      uint32_t leaf;
      if (thresh) {
        uint32_t rank = 0;
        for (uint32_t i = 0; i < ni; i++) { const int32_t ti = __builtin_amdgcn_readlane(t, (int)i); rank += (ti < t || (ti == t && i < lane)) ? 1u : 0u; }
        if (lane < ni) StS<int32_t>(LR + kWaSorted + 4 * rank, t);       // (the region doubles as the high mask words of the general shape)
        WaveSync();
        const uint32_t kk = min(lane, ni);
        const int32_t rep = ni == 0 ? 0 : (kk == 0 ? LdS<int32_t>(LR + kWaSorted) : (int32_t)((uint32_t)LdS<int32_t>(LR + kWaSorted + 4 * (kk - 1)) + 1u));
        leaf = WaveWalkLeaf(T, 0, chan, stream_id, y, rep);
        wc.thr = t;
      } else {
        const int pr = lane < ni ? (int)LdS<uint32_t>(LR + kWgProp + 4 * lane) : -1;
        G.thr = t;
        G.aW = pr == 7 || pr == 9 || pr == 10 ? 1 : 0;
        G.aN = pr == 6 || pr == 9 || pr == 12 || pr == 13 ? 1 : (pr == 11 ? -1 : 0);
        G.aNW = pr == 9 || pr == 10 ? -1 : (pr == 11 ? 1 : 0);
        G.aNE = pr == 12 ? -1 : 0;
        G.aNN = pr == 13 ? -1 : 0;
        G.aE = pr == 15 ? 1 : 0;
        G.aX = pr == 3 ? 1 : 0;
        G.c0 = 0;
        G.mlo = lane < nl ? LdS<uint32_t>(LR + kWaMask + 4 * lane) : 0u; G.mhi = lane < nl ? LdS<uint32_t>(LR + kWgMaskHi + 4 * lane) : 0u;
        G.wlo = lane < nl ? LdS<uint32_t>(LR + kWaWant + 4 * lane) : 1u; G.whi = lane < nl ? LdS<uint32_t>(LR + kWgWantHi + 4 * lane) : 0u;
        leaf = LdS<uint32_t>(LR + kWaLeaf + 4 * min(lane, nl - 1));
        G.is1 = (leaf & 0xFF) == 1; G.is5 = (leaf & 0xFF) == 5; G.is6 = (leaf & 0xFF) == 6;
      }
      wc.cluster = leaf >> 8;
      wc.abase = wide_off + ((wc.cluster << wc.la) << 3); wc.cbase = cut_off + ((wc.cluster << wc.la) << 1);
      WaveSync();
    }
    const uint32_t e0 = erows + (uint32_t)(y & 1) * rowb, e1 = erows + (uint32_t)((y & 1) ^ 1) * rowb;
    r.left = y ? __builtin_amdgcn_readlane(p1r[0], 0) : 0;
    r.Nprev = r.left;
    if (USE_WP) { r.teW = 0; r.teNprev = y ? __builtin_amdgcn_readlane(t1r[0], 0) : 0; r.e0prev = 0; r.e1x = LdS<int32_t>(e1); r.aprev = r.e1x; }
    int32_t c[4] = {0, 0, 0, 0}, tc[4] = {0, 0, 0, 0};
#pragma unroll
    for (int seg = 0; seg < 4; seg++) {
      if (seg < nseg) {
        const int x0 = seg * 64, n = min(64, w - x0);
        const bool last_seg = seg == nseg - 1;
        const int32_t nextp = seg < 3 ? __builtin_amdgcn_readlane(p1r[seg < 3 ? seg + 1 : 3], 0) : 0, nextt = seg < 3 ? __builtin_amdgcn_readlane(t1r[seg < 3 ? seg + 1 : 3], 0) : 0;
        const int32_t p1s = WaveShl1(p1r[seg], nextp), te1s = WaveShl1(t1r[seg], nextt);
        int32_t curv = 0, tecur = 0;
        const int nn = last_seg ? n - 1 : n;
#define JXL_GSAMPLE(R0, LA, GE, XL) WaveGenSample<R0, LA, USE_WP, GE>(bits, state, r, XL, (uint32_t)(x0 + (XL)), p1r[seg], p1s, p2r[seg], t1r[seg], te1s, curv, tecur, e1, e0, div_off, L, G, BG, wc)
#define JXL_GROW(R0, GE) do { for (int xl = 0; xl < nn; xl++) JXL_GSAMPLE(R0, false, GE, xl); if (last_seg) JXL_GSAMPLE(R0, true, GE, nn); } while (0)
        if (BIG) { if (y == 0) JXL_GROW(true, 2); else JXL_GROW(false, 2); }
        else if (USE_WP && thresh) { if (y == 0) JXL_GROW(true, 0); else JXL_GROW(false, 0); }
        else { if (y == 0) JXL_GROW(true, 1); else JXL_GROW(false, 1); }
#undef JXL_GROW
#undef JXL_GSAMPLE
        if ((int)lane < n) StG(p + x0 + (int)lane, curv);
        c[seg] = curv; tc[seg] = tecur;
      }
    }
#pragma unroll
    for (int seg = 0; seg < 4; seg++) { p2r[seg] = y == 0 ? c[seg] : p1r[seg]; p1r[seg] = c[seg]; t1r[seg] = tc[seg]; }
    if (USE_WP) WaveSync();      // (this row's error stores before the next row's reads)
  }
  if (r.toobig) { WaveSync(); return false; }
  state_io = state;
  const uint64_t endpos = bits.BitPos();
  br.Init(reinterpret_cast<const uint8_t*>(br.words), endpos, (uint64_t)br.wend * 4);
  WaveSync();
  return true;
}

// BALLOT: general trees are evaluated by the whole wavefront (see the "ballot" path below) — the Modular kernels; the LF kernel
// of the VarDCT path keeps the single-lane loops (register budget).
// BALLOT 0: the LF kernel of the VarDCT path; 1: the Modular kernels; 2: the Modular kernels' big-tree instantiations (DecodeChannelWaveGen<.., BIG>: 72 more registers)
template <int BALLOT = 0>
// (inlined: as a real call — tried in round 6 for the build time, LfDecodeKernel is 85 000 instructions with six copies of this — the second LF group of a wavefront
// failed with four groups per workgroup: JXL_COOP_NOINLINE keeps the experiment)
#ifdef JXL_COOP_NOINLINE
#define JXL_COOP_ATTR __noinline__
#else
#define JXL_COOP_ATTR __forceinline__
#endif
__device__ JXL_COOP_ATTR void DecodeChannelCoop(BitReaderP& br, uint32_t& state, const ModTables& T_in, const ModularCtx& mc_in, const ChannelDesc& ch_in, int chan) {
  const ModularCtx& mc = mc_in;
  const ChannelDesc& ch = ch_in;
  if (ch.w == 0 || ch.h == 0) return;
  const uint32_t lane = threadIdx.x & 63, wb = T_in.wb;
  ModTables T = T_in;
  if (!T.tree_in_lds && T.prune_cap) {
    // The tree does not fit the LDS whole (bench.jxl: 6643 nodes, most of them splits on the stream id): copy only what
    // this (stream, channel) can reach — static splits resolved, nodes re-indexed depth-first — into this wavefront's slice.
    if (lane == 0) {
      const uint32_t stack = wb + kWorkOff + 64;          // entries: source node, parent slot | child << 31
      int sp = 0, ok = 1;
      uint32_t count = 0;
      StS<uint32_t>(stack, 0u); StS<uint32_t>(stack + 4, 0xFFFFFFFFu); sp = 1;
      while (sp > 0) {
        sp--;
        uint32_t src = LdS<uint32_t>(stack + 8 * sp);
        const uint32_t link = LdS<uint32_t>(stack + 8 * sp + 4);
        uint4 v = LdG(reinterpret_cast<const uint4*>(T.tree_g + src));
        uint32_t guard = 0;
        while ((int32_t)v.x == 0 || (int32_t)v.x == 1) {   // static properties: channel index, stream id
          const int32_t pv = (int32_t)v.x == 0 ? chan : (int32_t)mc.stream_id;
          src = pv > (int32_t)v.y ? v.z : v.w;
          v = LdG(reinterpret_cast<const uint4*>(T.tree_g + src));
          if (++guard > 4096) { ok = 0; break; }
        }
        if (!ok || count >= T.prune_cap) { ok = 0; break; }
        const uint32_t dst = count++;
        if ((int32_t)v.x < 0) v.z = (v.z & 0xFF) | ((uint32_t)LdG(T.ctx_map_g + (v.z >> 8)) << 8);   // leaf: context -> cluster
        StS<uint4>(T.prune_off + dst * 16, v);
        if (link != 0xFFFFFFFFu) StS<uint32_t>(T.prune_off + (link & 0x7FFFFFFFu) * 16 + ((link >> 31) ? 12 : 8), dst);   // parent's b / a
        if ((int32_t)v.x >= 0) {
          if (sp + 2 > 60) { ok = 0; break; }
          StS<uint32_t>(stack + 8 * sp, v.w); StS<uint32_t>(stack + 8 * sp + 4, dst | 0x80000000u); sp++;
          StS<uint32_t>(stack + 8 * sp, v.z); StS<uint32_t>(stack + 8 * sp + 4, dst); sp++;
        }
      }
      StS<int>(wb + kWorkOff + 28, ok);
    }
    WaveSync();
    if (LdS<int>(wb + kWorkOff + 28)) { T.tree_in_lds = true; T.node_base = T.prune_off; }
    WaveSync();
  }
#ifndef JXL_NO_WAVE_LF
  // ---- small trees over cheap properties with simple leaves: the wave-wide decoder (every row of the channel, or none)
  if (T.tree_in_lds && !mc.slow && T.code.wide_off != kNotInLds && T.code.cfg_off != kNotInLds) {
    if (lane == 0) WaveAnalyse(T, 0, chan, (int32_t)mc.stream_id, 0, /*explore=*/true);
    WaveSync();
    const int wave_ok = LdS<int>(wb + kLutOff + kWaHdr), wave_has_y = LdS<int>(wb + kLutOff + kWaHdr + 24), wave_need_n = LdS<int>(wb + kLutOff + kWaHdr + 32);
    WaveSync();
    if (wave_ok && (!wave_need_n || ch.w <= 256)) { DecodeChannelWave(br, state, T, ch, 0, chan, (int32_t)mc.stream_id, wave_has_y != 0); return; }
    // ---- general trees (cjxl's lossless property set, weighted predictor): the wave-wide decoder's general form, rows up to 256 samples
    if (ch.w <= 256) {
      if (lane == 0) WaveAnalyseGen(T, 0, chan, (int32_t)mc.stream_id, 0, /*explore=*/true);
      WaveSync();
      const int gen_ok = LdS<int>(wb + kLutOff + kWaHdr), gen_has_y = LdS<int>(wb + kLutOff + kWaHdr + 24), gen_wp = LdS<int>(wb + kLutOff + kWaHdr + 36);
      WaveSync();
#ifdef JXL_WAVE_DEBUG
      if (lane == 0) printf("wave gen: stream %u chan %d w %d h %d ok %d ni %u nl %u wp %d wp_off %x simple_ok %d\n", mc.stream_id, chan, ch.w, ch.h, gen_ok, LdS<uint32_t>(wb + kLutOff + kWaHdr + 4), LdS<uint32_t>(wb + kLutOff + kWaHdr + 8), gen_wp, T.wp_off, wave_ok);
#endif
      if (gen_ok) {
        if (!gen_wp) { if (DecodeChannelWaveGen<false, false>(br, state, T, ch, mc.wp, chan, (int32_t)mc.stream_id, gen_has_y != 0)) return; }
        else if (T.wp_off != 0xFFFFFFFFu) { if (DecodeChannelWaveGen<true, false>(br, state, T, ch, mc.wp, chan, (int32_t)mc.stream_id, gen_has_y != 0)) return; }
      } else if (BALLOT == 2) {
        // ---- trees of hundreds of splits per stream (cjxl's lossless modes): the big-tree form — the Modular kernels only
        if (lane == 0) WaveAnalyseBig(T, wb + kWinOff, chan, (int32_t)mc.stream_id, 0, /*explore=*/true);
        WaveSync();
        const int big_ok = LdS<int>(wb + kLutOff + kWaHdr), big_has_y = LdS<int>(wb + kLutOff + kWaHdr + 24), big_wp = LdS<int>(wb + kLutOff + kWaHdr + 36);
        WaveSync();
#ifdef JXL_WAVE_DEBUG
        if (lane == 0) printf("wave big: stream %u chan %d w %d h %d ok %d wp %d cnt %u %u %u %u %u %u\n", mc.stream_id, chan, ch.w, ch.h, big_ok, big_wp, LdS<uint32_t>(wb + kLutOff + kWbCnt), LdS<uint32_t>(wb + kLutOff + kWbCnt + 4),
                              LdS<uint32_t>(wb + kLutOff + kWbCnt + 8), LdS<uint32_t>(wb + kLutOff + kWbCnt + 12), LdS<uint32_t>(wb + kLutOff + kWbCnt + 16), LdS<uint32_t>(wb + kLutOff + kWbCnt + 20));
#endif
        if (big_ok) {
          if (!big_wp) { if (DecodeChannelWaveGen<false, true>(br, state, T, ch, mc.wp, chan, (int32_t)mc.stream_id, big_has_y != 0)) return; }
          else if (T.wp_off != 0xFFFFFFFFu) { if (DecodeChannelWaveGen<true, true>(br, state, T, ch, mc.wp, chan, (int32_t)mc.stream_id, big_has_y != 0)) return; }
        }
      }
    }
  }
#endif
  // ---- lane 0: resolve static properties (channel, stream id) and analyse the remaining subtree
  if (lane == 0) {
    uint32_t pos = 0;
    TreeNode n = T.Node(0);
    while (n.prop == 0 || n.prop == 1) {
      const int32_t v = n.prop == 0 ? chan : (int32_t)mc.stream_id;
      pos = v > n.val ? n.a : n.b;
      n = T.Node(pos);
    }
    int mode = 1, prop = -1, count = 0, wide = 0;
    uint32_t ni = 0;  // inner nodes seen; the first 64 constants go to the (not yet built) LUT region for the wave-wide decoder
    int upred = -1;   // predictor shared by all leaves with offset 0 / multiplier 1 (-2: not uniform / not simple)
    int sp = 0;       // iterative DFS with a bounded stack at kWorkOff + 64
    StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)pos);
    while (sp > 0 && mode == 1) {
      const TreeNode m = T.Node((uint32_t)LdS<int>(wb + kWorkOff + 64 + 4 * --sp));
      if (++count > 600) { mode = 0; break; }
      if (m.prop < 0) {
        if ((m.a & 0xFF) == 6) mode = 0;                                     // weighted predictor: general path
        if (m.val != 0 || m.b != 1) upred = -2;
        else if (upred == -1) upred = (int)(m.a & 0xFF);
        else if (upred != (int)(m.a & 0xFF)) upred = -2;
        continue;
      }
      if (m.prop == 15 || m.prop >= 16 || m.prop <= 1) { mode = 0; break; }
      if (prop < 0) prop = m.prop; else if (prop != m.prop) { mode = 0; break; }
      if (sp + 2 > 200) { mode = 0; break; }
      if (m.val < -512 || m.val > 510) wide = 1;      // split outside the LUT's range: such property values walk the subtree
      if (ni < 64) StS<int32_t>(wb + kLutOff + 4 * ni, m.val);
      ni++;
      StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)m.a); StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)m.b);
    }
    if (!T.tree_in_lds) mode = 0;
    // whether the channel's subtree looks at the weighted predictor at all (a leaf with predictor 6, a split on property 15): its state is
    // per channel, so a channel that never does skips the arithmetic — and the state rows, which for the widest channels (the block-info rows
    // of an LF group: up to 65 536 samples) the host only provides when the subtree can get there (decoder.cc LfChannelUsesWp)
    int sub_wp = 0;
    if (mc.uses_wp && mode == 0) {
      int visited = 0;
      sp = 0;
      StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)pos);
      while (sp > 0 && !sub_wp) {
        const TreeNode m = T.Node((uint32_t)LdS<int>(wb + kWorkOff + 64 + 4 * --sp));
        if (++visited > 8192) { sub_wp = 1; break; }
        if (m.prop < 0) { if ((m.a & 0xFF) == 6) sub_wp = 1; continue; }
        if (m.prop == 15) { sub_wp = 1; break; }
        if (m.prop == 0 || m.prop == 1) { StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)((m.prop == 0 ? chan : (int32_t)mc.stream_id) > m.val ? m.a : m.b)); continue; }
        if (sp + 2 > 200) { sub_wp = 1; break; }
        StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)m.a); StS<int>(wb + kWorkOff + 64 + 4 * sp++, (int)m.b);
      }
    }
    StS<int>(wb + kWorkOff + 56, sub_wp);
    StS<int>(wb + kWorkOff + 0, mode); StS<int>(wb + kWorkOff + 4, prop); StS<int>(wb + kWorkOff + 8, (int)pos); StS<int>(wb + kWorkOff + 12, upred);
    StS<int>(wb + kWorkOff + 20, wide);
    StS<uint32_t>(wb + kWorkOff + 60, ni);
  }
  WaveSync();
  const int mode = mc.slow ? 0 : LdS<int>(wb + kWorkOff + 0), prop = LdS<int>(wb + kWorkOff + 4);   // prefix / LZ77 streams: general loop only
  const uint32_t subroot = (uint32_t)LdS<int>(wb + kWorkOff + 8);
  const int upred = LdS<int>(wb + kWorkOff + 12);
  const uint32_t wide_subroot = LdS<int>(wb + kWorkOff + 20) ? subroot : 0xFFFFFFFFu;
  // fast rows: leaves are (predictor p, offset 0, multiplier 1) with p in {0 zero, 1 W, 5 gradient}; the LUT then maps
  // the property value straight to the cluster
  const bool need_n = upred == 5 || prop == 9;    // previous row needed
  const bool fast = mode == 1 && (upred == 0 || upred == 1 || upred == 5) && (prop < 0 || prop == 2 || prop == 9) &&
                    T.code.cfg_off != kNotInLds && T.code.alias_off != kNotInLds && ((uint32_t)ch.w <= kRowMax || !need_n);
  if (mode == 1) {
    for (int i = (int)lane; i < 1024; i += 64) {
      const int32_t v = i - 512;
      uint32_t pos = subroot;
      TreeNode n = T.Node(pos);
      while (n.prop >= 0) { pos = v > n.val ? n.a : n.b; n = T.Node(pos); }
      StS<uint16_t>(wb + kLutOff + 2 * i, fast ? (uint16_t)(n.a >> 8) : (uint16_t)pos);   // fast: cluster, else leaf node index
    }
  }
  WaveSync();
  if (fast) {
    // ---- LDS-only serial loop.  Lane 0 decodes 256 samples at a time from a bit-stream window in LDS, reading the
    // previous row from LDS and writing into LDS; the other lanes load the window and flush finished rows / chunks with
    // coalesced global accesses.
    const int w = ch.w, h = ch.h;
    const bool row_in_lds = (uint32_t)w <= kRowMax;
    const uint32_t cfg_off = T.code.cfg_off, alias_off = T.code.alias_off, la = T.code.log_alpha;
    const uint32_t cfg_uniform = T.code.cfg_uniform;
    const uint32_t wend = br.wend;
    BitReaderW bw;
    bw.wpos = 0; bw.win_base = 0; bw.buf = 0; bw.avail = 0; bw.win_off = wb + kWinOff;
    uint32_t skip_bits = 0;
    if (lane == 0) { const uint64_t bp = br.BitPos(); bw.wpos = (uint32_t)(bp >> 5); skip_bits = (uint32_t)(bp & 31); }
    int32_t left = 0, nw = 0;
    uint32_t cur = wb + kRowOff, prev = wb + kRowOff + kRowMax * 4;
    for (int y = 0; y < h; y++) {
      int32_t* p = ch.data + (size_t)y * ch.stride;
      uint32_t cl_row = 0;
      if (prop != 9) {
        const int32_t v0 = prop == 2 ? y : 0;
        const int32_t v = v0 > 511 ? 511 : v0;
        cl_row = (wide_subroot != 0xFFFFFFFFu && v != v0) ? WalkCluster(T.node_base, wide_subroot, v0) : LdS<uint16_t>(wb + kLutOff + 2 * (uint32_t)(v + 512));
      }
      for (int x0 = 0; x0 < w; x0 += 256) {
        if (lane == 0) StS<uint32_t>(wb + kWorkOff + 16, bw.wpos);
        WaveSync();
        const uint32_t wbase = LdS<uint32_t>(wb + kWorkOff + 16);
        for (uint32_t i = lane; i < kWinWords; i += 64) StS<uint32_t>(wb + kWinOff + i * 4, wbase + i < wend ? LdG(br.words + wbase + i) : 0u);
        WaveSync();
        if (lane == 0) {
          bw.win_base = wbase;
          if (skip_bits != 0xFFFFFFFFu) {   // first chunk of the channel: establish the bit buffer
            bw.buf = 0; bw.avail = 0;
            bw.Refill(); bw.buf >>= skip_bits; bw.avail -= (int)skip_bits;
            skip_bits = 0xFFFFFFFFu;
          }
        }
        {
          const int x1 = min(w, x0 + 256);
          const uint32_t obase = row_in_lds ? cur : wb + kChunkOff - (uint32_t)x0 * 4;
          ChunkState st;
          st.bw = bw; st.state = state; st.left = left; st.nw = nw;
          const uint32_t lut_off = wb + kLutOff, first_off = wb + kWorkOff + 24;
          const int rm = y == 0 ? 0 : (need_n ? 1 : 2);
#define JXL_CHUNK_ARGS st, x0, x1, prev, obase, lut_off, first_off, cl_row, cfg_off, cfg_uniform, alias_off, la, wide_subroot, T.node_base
#define JXL_DISPATCH                                                                              \
          if (prop == 9) {                                                                        \
            if (upred == 5) { if (rm == 0) JXL_CHUNK(0, true, 5); else JXL_CHUNK(1, true, 5); }   \
            else if (upred == 1) { if (rm == 0) JXL_CHUNK(0, true, 1); else JXL_CHUNK(1, true, 1); } \
            else { if (rm == 0) JXL_CHUNK(0, true, 0); else JXL_CHUNK(1, true, 0); }              \
          } else {                                                                                \
            if (upred == 5) { if (rm == 0) JXL_CHUNK(0, false, 5); else JXL_CHUNK(1, false, 5); } \
            else if (upred == 1) { if (rm == 0) JXL_CHUNK(0, false, 1); else JXL_CHUNK(2, false, 1); } \
            else { if (rm == 0) JXL_CHUNK(0, false, 0); else JXL_CHUNK(2, false, 0); }            \
          }
          if (lane == 0) {
#define JXL_CHUNK(RM, P9, UP) do { if (cfg_uniform != 0xFFFFFFFFu) DecodeChunkLds<RM, P9, UP, true>(JXL_CHUNK_ARGS); else DecodeChunkLds<RM, P9, UP, false>(JXL_CHUNK_ARGS); } while (0)
            JXL_DISPATCH
#undef JXL_CHUNK
            bw = st.bw; state = st.state; left = st.left; nw = st.nw;
          }
#undef JXL_DISPATCH
#undef JXL_CHUNK_ARGS
        }
        WaveSync();
        if (!row_in_lds) {
          const int n = min(256, w - x0);
          for (int i = (int)lane; i < n; i += 64) StG(p + x0 + i, LdS<int32_t>(wb + kChunkOff + 4 * i));
        }
      }
      if (row_in_lds) {
        for (int i = (int)lane; i < w; i += 64) StG(p + i, LdS<int32_t>(cur + 4 * i));
        const uint32_t t = cur; cur = prev; prev = t;
      }
    }
    // hand the bit position back to the generic reader
    if (lane == 0) StS<uint64_t>(wb + kWorkOff + 32, bw.BitPos());
    WaveSync();
    const uint64_t endpos = LdS<uint64_t>(wb + kWorkOff + 32);
    br.Init(reinterpret_cast<const uint8_t*>(br.words), endpos, (uint64_t)br.wend * 4);
    WaveSync();
    return;
  }
  const bool use_wp = mc.uses_wp != 0 && mode == 0 && LdS<int>(wb + kWorkOff + 56) != 0;
  const bool wp_in_lds = use_wp && T.wp_off != 0xFFFFFFFFu && ch.w <= kWpLdsMaxW;
  if (use_wp && !wp_in_lds && 10ull * ((uint64_t)ch.w + 2) > mc.wp_scratch_ints) {     // (a tree too large to survey: no state rows of that width)
    if (lane == 0 && mc.status) atomicOr(mc.status, kErrUnsupported);
    return;
  }
  WPStateLds wpl;
  if (wp_in_lds) { wpl.Init(T.wp_off, T.wp_off + kWpLdsBytes - 256, ch.w, lane, mc.narrow_wp != 0); WaveSync(); }
  // ---- general trees / predictors with everything the sample loop touches in LDS (no vector-memory instruction, hence no
  // vmcnt wait, per sample): tree (whole or pruned), alias tables, bit-stream window, the three sample rows the properties
  // and predictors read, the WP state.  Rows up to kRowMax samples; wider channels take the loop below.
  const bool lds_generic = !mc.slow && mc.max_prop < 16 && T.tree_in_lds && T.code.cfg_off != kNotInLds && (T.code.alias_off != kNotInLds || T.code.wide_off != kNotInLds) && (uint32_t)ch.w <= kRowMax && (!use_wp || wp_in_lds);
  // ---- "ballot" path: the whole wavefront decodes the channel together.  Every value of the serial chain (neighbours, ANS state,
  // bit buffer, weighted-predictor arithmetic) is computed redundantly by all 64 lanes — that costs nothing, a wavefront
  // instruction takes the same time for one active lane as for 64 — and the part that used to be a pointer chase through LDS is
  // spread over the lanes: lane j owns inner node j of the channel's subtree (at most 64 inner nodes and 64 leaves after the
  // static splits are resolved), evaluates that node's property and comparison, one ballot yields all decisions of the tree as
  // a 64-bit scalar, and the walk from the root is a few scalar bit tests with the child tables read across lanes
  // (v_readlane) — no LDS round trip per tree level (the single-lane loop paid two).  Channels whose subtree never looks at the
  // weighted predictor (no predictor 6 leaf, no property 15 split) skip its arithmetic altogether.
#ifndef JXL_NO_BALLOT
  if (BALLOT && lds_generic && mode == 0) {
    // breadth-first numbering of the channel's subtree (static splits resolved): inner node i -> lane i % 64, register row i / 64
    const uint32_t qi_off = wb + kLutOff, ql_off = wb + kLutOff + 1024, pair_off = wb + kWinOff;   // u16[512], u16[512], u32[512] (the window is loaded later)
    if (lane == 0) {
      uint32_t ni = 0, nl = 0;
      int ok = 1;
      auto resolve = [&](uint32_t pos) {     // follow static splits (channel index, stream id)
        TreeNode n = T.Node(pos);
        uint32_t guard = 0;
        while ((n.prop == 0 || n.prop == 1) && ++guard < 4096) { pos = (n.prop == 0 ? chan : (int32_t)mc.stream_id) > n.val ? n.a : n.b; n = T.Node(pos); }
        return pos;
      };
      const uint32_t root = resolve(subroot);
      uint32_t root_code;
      if (root > 0xFFFFu) ok = 0;
      if (T.Node(root).prop < 0) { StS<uint16_t>(ql_off, (uint16_t)root); nl = 1; root_code = 0x8000; }
      else { StS<uint16_t>(qi_off, (uint16_t)root); ni = 1; root_code = 0; }
      for (uint32_t i = 0; i < ni && ok; i++) {
        const TreeNode n = T.Node(LdS<uint16_t>(qi_off + 2 * i));
        uint32_t codes[2];
        for (int k = 0; k < 2; k++) {
          const uint32_t c = resolve(k == 0 ? n.a : n.b);
          if (c > 0xFFFFu) { ok = 0; break; }
          if (T.Node(c).prop < 0) { if (nl >= kBallotMax) { ok = 0; break; } StS<uint16_t>(ql_off + 2 * nl, (uint16_t)c); codes[k] = 0x8000 | nl++; }
          else { if (ni >= kBallotMax) { ok = 0; break; } StS<uint16_t>(qi_off + 2 * ni, (uint16_t)c); codes[k] = ni++; }
        }
        if (ok) StS<uint32_t>(pair_off + 4 * i, codes[0] | (codes[1] << 16));
      }
      StS<int>(wb + kWorkOff + 40, ok); StS<uint32_t>(wb + kWorkOff + 44, ni); StS<uint32_t>(wb + kWorkOff + 48, nl); StS<uint32_t>(wb + kWorkOff + 52, root_code);
    }
    WaveSync();
    if (LdS<int>(wb + kWorkOff + 40)) {
      const uint32_t ni = LdS<uint32_t>(wb + kWorkOff + 44), nl = LdS<uint32_t>(wb + kWorkOff + 48);
      const uint32_t rows = (max(ni, nl) + 63) / 64;
      const BallotArgs ba{qi_off, ql_off, pair_off, ni, nl, Uniform(LdS<uint32_t>(wb + kWorkOff + 52)), use_wp};
      if (rows <= 1) DecodeRowsBallot<1>(br, state, T, mc, ch, chan, wpl, ba);
      else if (rows <= 2) DecodeRowsBallot<2>(br, state, T, mc, ch, chan, wpl, ba);
      else if (rows <= 4) DecodeRowsBallot<4>(br, state, T, mc, ch, chan, wpl, ba);
      else DecodeRowsBallot<8>(br, state, T, mc, ch, chan, wpl, ba);
      return;
    }
  }
#endif
  if (lds_generic) {
    const int w = ch.w, h = ch.h;
    const uint32_t cfg_off = T.code.cfg_off, alias_off = T.code.alias_off, la = T.code.log_alpha;
    const uint32_t wend = br.wend;
    BitReaderW bw;
    bw.wpos = 0; bw.win_base = 0; bw.buf = 0; bw.avail = 0; bw.win_off = wb + kWinOff;
    uint32_t skip_bits = 0;
    if (lane == 0) { const uint64_t bp = br.BitPos(); bw.wpos = (uint32_t)(bp >> 5); skip_bits = (uint32_t)(bp & 31); }
    uint32_t cur = wb + kRowOff, prev = wb + kRowOff + kRowMax * 4, prev2 = wb + kChunkOff;   // three row buffers, rotated
    for (int y = 0; y < h; y++) {
      int32_t* p = ch.data + (size_t)y * ch.stride;
      if (lane == 0) StS<uint32_t>(wb + kWorkOff + 16, bw.wpos);
      WaveSync();
      const uint32_t wbase = LdS<uint32_t>(wb + kWorkOff + 16);
      for (uint32_t i = lane; i < kWinWords; i += 64) StS<uint32_t>(wb + kWinOff + i * 4, wbase + i < wend ? LdG(br.words + wbase + i) : 0u);
      WaveSync();
      if (lane == 0) {
        bw.win_base = wbase;
        if (skip_bits != 0xFFFFFFFFu) { bw.buf = 0; bw.avail = 0; bw.Refill(); bw.buf >>= skip_bits; bw.avail -= (int)skip_bits; skip_bits = 0xFFFFFFFFu; }
        int32_t left = 0, left2 = 0, prev9 = 0;
        int32_t up0 = 0, up1 = 0, up2 = 0, up3 = 0;
        if (y > 0) { up1 = LdS<int32_t>(prev); up2 = w > 1 ? LdS<int32_t>(prev + 4) : 0; up3 = w > 2 ? LdS<int32_t>(prev + 8) : 0; }
        for (int x = 0; x < w; x++) {
          const int32_t up4 = (y > 0 && x + 3 < w) ? LdS<int32_t>(prev + 4 * (x + 3)) : 0;
          const int32_t W = x ? left : (y ? up1 : 0);
          const int32_t N = y ? up1 : W;
          const int32_t NW = (x && y) ? up0 : W;
          const int32_t NE = (x + 1 < w && y) ? up2 : N;
          const int32_t WW = x > 1 ? left2 : W;
          const int32_t NN = y > 1 ? LdS<int32_t>(prev2 + 4 * x) : N;
          const int32_t NEE = (x + 2 < w && y) ? up3 : NE;
          int64_t wp_pred = 0;
          int32_t wp_err = 0;
          if (use_wp) wp_pred = wpl.Predict(mc.wp, x, y, N, W, NE, NW, NN, &wp_err);
          TreeNode n;
          if (mode == 1) {
            const int32_t v0 = prop < 0 ? 0 : PropValue(prop, chan, mc.stream_id, x, y, W, N, NW, NE, NN, WW, prev9, 0);
            const int32_t v = v0 < -512 ? -512 : (v0 > 511 ? 511 : v0);
            if (wide_subroot != 0xFFFFFFFFu && v != v0) {      // a split beyond the LUT's range and a value out there: walk the subtree
              n = T.Node(wide_subroot);
              while (n.prop >= 0) n = T.Node(v0 > n.val ? n.a : n.b);
            } else n = T.Node(LdS<uint16_t>(wb + kLutOff + 2 * (uint32_t)(v + 512)));
          } else {
            // all 16 properties of the sample once, into LDS; every tree level then costs a node read and one indexed read
            // instead of a 16-way select over recomputed values
            const uint32_t pv = wb + kWorkOff + 896;
            const int32_t grad = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
            StS<int4>(pv, make_int4(chan, (int32_t)mc.stream_id, y, x));
            StS<int4>(pv + 16, make_int4(N < 0 ? (int32_t)(0u - (uint32_t)N) : N, W < 0 ? (int32_t)(0u - (uint32_t)W) : W, N, W));
            StS<int4>(pv + 32, make_int4((int32_t)((uint32_t)W - (uint32_t)prev9), grad, (int32_t)((uint32_t)W - (uint32_t)NW), (int32_t)((uint32_t)NW - (uint32_t)N)));
            StS<int4>(pv + 48, make_int4((int32_t)((uint32_t)N - (uint32_t)NE), (int32_t)((uint32_t)N - (uint32_t)NN), (int32_t)((uint32_t)W - (uint32_t)WW), wp_err));
            uint32_t pos = subroot;
            n = T.Node(pos);
            while (n.prop >= 0) {
              const int32_t v = LdS<int32_t>(pv + 4 * (uint32_t)(n.prop & 15));
              pos = v > n.val ? n.a : n.b;
              n = T.Node(pos);
            }
          }
          prev9 = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
          const uint32_t cluster = n.a >> 8;
          const int32_t guess = Predict(n.a & 0xFF, W, N, NW, NE, NN, WW, NEE, wp_pred);
          // ANS symbol + hybrid integer out of LDS (same as DecodeChunkLds)
          const uint32_t res = state & 0xFFF;
          const uint32_t i = res >> (12 - la), pos_ = res & ((1u << (12 - la)) - 1);
          const uint64_t e = LdAliasAt(alias_off, T.code.wide_off, T.code.cut_off, (cluster << la) + i);
          const uint32_t cfg = LdS<uint32_t>(cfg_off + cluster * 4);
          const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
          const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
          const bool hit = pos_ >= cutoff;
          uint32_t tok = hit ? right : i;
          state = (hit ? freq1 : freq0) * (state >> 12) + (hit ? offs1 + pos_ : pos_);
          if (state < (1u << 16)) state = (state << 16) | bw.Read(16);
          const uint32_t split_exp = cfg & 0xFF;
          if (tok >= (1u << split_exp)) {
            const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
            const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - (1u << split_exp)) >> (msb + lsb))) & 31;
            const uint32_t low = tok & ((1u << lsb) - 1);
            tok >>= lsb;
            const uint32_t bits = nbits ? bw.Read((int)nbits) : 0;
            const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
            tok = (((hi << nbits) | bits) << lsb) | low;
          }
          const int32_t val = (int32_t)((uint32_t)UnpackSigned(tok) * n.b + (uint32_t)n.val + (uint32_t)guess);
          StS<int32_t>(cur + 4 * x, val);
          if (use_wp) wpl.Update(val, x, y);
          left2 = left; left = val;
          up0 = up1; up1 = up2; up2 = up3; up3 = up4;
        }
      }
      WaveSync();
      for (int i = (int)lane; i < w; i += 64) StG(p + i, LdS<int32_t>(cur + 4 * i));
      const uint32_t t = prev2; prev2 = prev; prev = cur; cur = t;
    }
    if (lane == 0) StS<uint64_t>(wb + kWorkOff + 32, bw.BitPos());
    WaveSync();
    const uint64_t endpos = LdS<uint64_t>(wb + kWorkOff + 32);
    br.Init(reinterpret_cast<const uint8_t*>(br.words), endpos, (uint64_t)br.wend * 4);
    WaveSync();
    return;
  }
  if (lane == 0) {
    const int w = ch.w, h = ch.h;
    const FastCode& code = T.code;
    WPState wps;
    if (use_wp && !wp_in_lds) wps.Init(mc.wp_scratch, w);
    for (int y = 0; y < h; y++) {
      int32_t* p = ch.data + (size_t)y * ch.stride;
      const int32_t* pn = p - ch.stride;
      const int32_t* pnn = pn - ch.stride;
      // sliding window over the row above (loads issued ahead of use) and the row above that
      int32_t up1 = 0, up2 = 0, up3 = 0, up0 = 0, nn1 = 0, nn2 = 0;
      if (y > 0) { up1 = LdG(pn); up2 = w > 1 ? LdG(pn + 1) : 0; up3 = w > 2 ? LdG(pn + 2) : 0; }
      if (y > 1) { nn1 = LdG(pnn); nn2 = w > 1 ? LdG(pnn + 1) : 0; }
      int32_t left = 0, left2 = 0, prev9 = 0;
      for (int x = 0; x < w; x++) {
        const int32_t up4 = (y > 0 && x + 3 < w) ? LdG(pn + x + 3) : 0;     // prefetch for iteration x+1
        const int32_t nn3 = (y > 1 && x + 2 < w) ? LdG(pnn + x + 2) : 0;
        const int32_t W = x ? left : (y ? up1 : 0);
        const int32_t N = y ? up1 : W;
        const int32_t NW = (x && y) ? up0 : W;
        const int32_t NE = (x + 1 < w && y) ? up2 : N;
        const int32_t WW = x > 1 ? left2 : W;
        const int32_t NN = y > 1 ? nn1 : N;
        const int32_t NEE = (x + 2 < w && y) ? up3 : NE;
        int64_t wp_pred = 0;
        int32_t wp_err = 0;
        if (use_wp) wp_pred = wp_in_lds ? wpl.Predict(mc.wp, x, y, N, W, NE, NW, NN, &wp_err) : wps.Predict(mc.wp, x, y, N, W, NE, NW, NN, &wp_err);
        TreeNode n;
        if (mode == 1) {
          const int32_t v0 = prop < 0 ? 0 : PropValue(prop, chan, mc.stream_id, x, y, W, N, NW, NE, NN, WW, prev9, 0);
          const int32_t v = v0 < -512 ? -512 : (v0 > 511 ? 511 : v0);
          if (wide_subroot != 0xFFFFFFFFu && v != v0) {      // a split beyond the LUT's range and a value out there: walk the subtree
            n = T.Node(wide_subroot);
            while (n.prop >= 0) n = T.Node(v0 > n.val ? n.a : n.b);
          } else n = T.Node(LdS<uint16_t>(wb + kLutOff + 2 * (uint32_t)(v + 512)));
        } else {
          uint32_t pos = subroot;
          n = T.Node(pos);
          while (n.prop >= 0) {
            const int32_t v = n.prop < 16 ? PropValue(n.prop, chan, mc.stream_id, x, y, W, N, NW, NE, NN, WW, prev9, wp_err) : RefPropValue(mc.refs, n.prop, x, y);
            pos = v > n.val ? n.a : n.b;
            n = T.Node(pos);
          }
        }
        prev9 = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
        const uint32_t predictor = n.a & 0xFF;
        const uint32_t cluster = T.tree_in_lds ? (n.a >> 8) : code.Cluster(n.a >> 8);
        const int32_t guess = Predict(predictor, W, N, NW, NE, NN, WW, NEE, wp_pred);
        uint32_t tok;
        if (!mc.slow) tok = FastHybrid(br, state, code, cluster);
        else {
          // prefix codes / LZ77 (dec_ans.h): general symbol reader over the tables in global memory
          const DevCode& dc = *mc.code;
          AnsReader ans; ans.state = state;
          if (!dc.lz77) tok = HybridFromToken(br, dc.cfg[cluster], ReadSymbol(br, ans, dc, cluster));
          else tok = Lz77Read(br, *mc.lz, cluster, (uint32_t)dc.ctx_map[dc.num_ctx], dc.lz_min_symbol, dc.lz_min_length, dc.lz_len_cfg,
                              [](uint32_t c) { return c; }, [&](uint32_t cl) { return ReadSymbol(br, ans, dc, cl); }, [&](uint32_t cl) { return dc.cfg[cl]; });
          state = ans.state;
        }
        const int32_t val = (int32_t)((uint32_t)UnpackSigned(tok) * n.b + (uint32_t)n.val + (uint32_t)guess);
        StG(p + x, val);
        if (use_wp) { if (wp_in_lds) wpl.Update(val, x, y); else wps.Update(val, x, y); }
        left2 = left; left = val;
        up0 = up1; up1 = up2; up2 = up3; up3 = up4;
        nn1 = nn2; nn2 = nn3;
      }
    }
  }
  WaveSync();
}

// Stages tree + code into the shared part of the LDS (all threads of the block; one block barrier).
__device__ void StageModular(const TreeNode* tree, uint32_t num_tree_nodes, const DevCode& code, ModTables& T, uint32_t tree_cap, uint32_t lds_bytes) {
  const uint32_t tree_off = (blockDim.x >> 6) * kWaveLds;   // shared part starts after the per-wavefront regions
  const uint32_t code_base = tree_off + tree_cap * 16;
  const uint32_t budget = lds_bytes > code_base ? lds_bytes - code_base : 0;
  StageCode(code, T.code, code_base, budget, /*with_ctx_map=*/false, /*with_wide=*/true);
  T.tree_g = tree;
  T.tree_cap = tree_cap;
  T.tree_in_lds = num_tree_nodes <= tree_cap;
  T.node_base = tree_off;
  T.ctx_map_g = code.ctx_map;
  {  // larger trees: every wavefront of the workgroup gets a slice of the region for a pruned subtree
    const uint32_t nw = blockDim.x >> 6;
    T.prune_cap = T.tree_in_lds ? 0 : tree_cap / nw;
    T.prune_off = tree_off + (threadIdx.x >> 6) * T.prune_cap * 16;
  }
  T.wb = (threadIdx.x >> 6) * kWaveLds;
  T.wp_off = 0xFFFFFFFFu;
  if (T.tree_in_lds) {
    for (uint32_t i = threadIdx.x; i < num_tree_nodes; i += blockDim.x) {
      uint4 v = LdG(reinterpret_cast<const uint4*>(tree + i));
      if ((int32_t)v.x < 0) v.z = (v.z & 0xFF) | ((uint32_t)LdG(code.ctx_map + (v.z >> 8)) << 8);   // leaf: context -> cluster
      StS<uint4>(tree_off + i * 16, v);
    }
  }
  __syncthreads();
}

__device__ __forceinline__ int32_t* ModPlane(const FrameDev& f, const ModChanDev& c) { return (int32_t*)(f.mod_base + c.off); }

// Rectangle of global-image channel c inside the section unit (x0, y0, dim) for the shift range [min_shift, max_shift]
// (dec_modular.cc DecodeGroup); false: the channel is not part of that sub-stream.
__device__ __forceinline__ bool ModUnitRect(const FrameDev& f, uint32_t c, uint32_t x0, uint32_t y0, uint32_t dim, int min_shift, int max_shift, ChannelDesc* d) {
  const ModChanDev m = f.mod_chan[c];
  if (m.w == 0 || m.h == 0) return false;
  const int shift = min(m.hshift, m.vshift);
  if (shift < min_shift || shift > max_shift) return false;
  const uint32_t rx = x0 >> m.hshift, ry = y0 >> m.vshift;
  if (rx >= m.w || ry >= m.h) return false;
  const uint32_t rw = min(dim >> m.hshift, m.w - rx), rh = min(dim >> m.vshift, m.h - ry);
  if (rw == 0 || rh == 0) return false;
  d->data = ModPlane(f, m) + (size_t)ry * m.w + rx; d->w = (int)rw; d->h = (int)rh; d->stride = (int)m.w; d->hs = m.hshift; d->vs = m.vshift;
  return true;
}

// =====================================================================================================================
// K_lf: one wavefront per LF group (four per workgroup, sharing the tables) — LF coefficients (3 channels, order Y,X,B)
// + HF metadata, then varblock placement
// =====================================================================================================================
// Entropy decode of one LF group by one wavefront (DecodeChannelCoop): three LF coefficient channels into the quantised LF planes, four
// HF-metadata channels and the block count (scratch[1]) into the group's scratch.  What follows — chroma-from-luma maps, varblock
// placement, block-info words — is LfPlaceKernel's, for this kernel's frames and the SIMT kernel's alike.
__device__ __forceinline__ void LfDecodeGroup(const FrameDev& f, const uint32_t g, ModTables& T, int& s_fail, GroupHeaderD& s_gh, uint32_t* s_u) {
  const uint32_t lane = threadIdx.x & 63;
  const uint32_t gx = g % f.xlfgroups, gy = g / f.xlfgroups;
  const uint32_t bx0 = gx * 256, by0 = gy * 256;
  const uint32_t gbw = min(256u, f.bw - bx0), gbh = min(256u, f.bh - by0);
  BitReaderP br;
  const uint64_t sec_end = f.single_section ? f.cs_size : f.sec_off[1 + g] + f.sec_size[1 + g];
  if (f.single_section) br.Init(f.cs, f.mod_nchan ? f.stream_end_bitpos[1] : f.lf_start_bitpos, f.cs_size);   // (after the global Modular stream, if any)
  else br.Init(f.cs, f.sec_off[1 + g] * 8, sec_end);
  const uint64_t limit = sec_end * 8;
  if (lane == 0) s_fail = 0;
  WaveSync();

  ModularCtx mc;
  mc.tree = f.tree; mc.code = &f.mod_code; mc.uses_wp = f.uses_wp;
  mc.wp_scratch = f.wp_scratch + (uint64_t)g * f.wp_scratch_stride; mc.wp_scratch_ints = f.wp_scratch_stride; mc.status = f.status;
  mc.slow = f.mod_code.use_prefix || f.mod_code.lz77;       // prefix-coded LF streams (cjxl's fast efforts), LZ77 (its slowest): the general symbol reader
  // LZ77 state in LDS (the rare path must not cost the common one registers); one window per LF group: its two streams — LF coefficients,
  // HF metadata — use it one after the other
  __shared__ Lz77State s_lz[kLfDecWaves];
  Lz77State& lz = s_lz[(threadIdx.x >> 6) % kLfDecWaves];
  if (f.mod_code.lz77) {
    if (!f.lz_window) { if (lane == 0) SetError(f, kErrUnsupported); return; }
    if (lane == 0) lz.Init(f.lz_window + (uint64_t)(f.lz_lf_base + g) * Lz77State::kWindow, gbw);   // dist_multiplier: the widest channel of the stream (modular/encoding/encoding.cc)
    mc.lz = &lz;
  }
  // previous-channel properties (16 + 4 r + k; cjxl -E): the earlier channels of the stream with the same size, nearest first (encoding.cc
  // PrecomputeReferences) — kept in LDS, built by lane 0 in front of every channel
  __shared__ ModRefs s_refs[kLfDecWaves];
  __shared__ ChannelDesc s_done[kLfDecWaves][4];
  const uint32_t wslot = (threadIdx.x >> 6) % kLfDecWaves;
  const bool with_refs = f.tree_max_prop >= 16;
  mc.max_prop = f.tree_max_prop;
  if (with_refs) mc.refs = &s_refs[wslot];
  auto before_channel = [&](const ChannelDesc& chd, int k) {      // k: index of the channel in its stream
    if (!with_refs) return;
    if (lane == 0) {
      ModRefs& r = s_refs[wslot];
      int n = 0;
      for (int j = k - 1; j >= 0 && n < kMaxModRefs; j--) {
        const ChannelDesc& d = s_done[wslot][j];
        if (d.w != chd.w || d.h != chd.h) continue;
        r.data[n] = d.data; r.stride[n] = d.stride; n++;
      }
      r.n = n;
      s_done[wslot][k] = chd;
    }
    WaveSync();
  };
  uint32_t state = 0;
  int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
  // ---- LF coefficients (not in the stream of a frame that takes its LF image from an LF frame: frame_header.cc kUseDcFrame)
  const bool has_lf = !f.use_lf_frame;
  if (lane == 0 && has_lf) {
    s_u[0] = br.Read(2);  // extra_precision
    BitReader tmp;        // GroupHeader parsing reuses the generic reader type: re-sync positions around it
    tmp.Init(f.cs, br.BitPos(), f.cs_size);
    if (!ReadGroupHeader(tmp, s_gh) || !s_gh.use_global_tree || s_gh.ntransforms != 0) { SetError(f, kErrUnsupported); s_fail = 1; }
    br.Init(f.cs, tmp.BitPos(), sec_end);
    state = f.mod_code.use_prefix ? 0x130000u : br.Read(32);     // (prefix codes carry no ANS state)
    scratch[0] = (int32_t)s_u[0];
    scratch[1] = 0;
  }
  if (lane == 0 && !has_lf) { scratch[0] = 0; scratch[1] = 0; state = 0x130000u; }
  WaveSync();
  if (s_fail) return;
  mc.wp = s_gh.wp; mc.stream_id = 1 + g;
  if (has_lf) {
    const int chan_to_plane[3] = {1, 0, 2};  // stream channel order is Y, X, B
    for (int c = 0; c < 3; c++) {
      ChannelDesc ch;
      const int pl = chan_to_plane[c];       // (subsampled channels: their own, smaller grid — dec_modular.cc DecodeVarDCTDC)
      ch.data = f.lfq[pl] + (size_t)(by0 >> f.vs[pl]) * f.bw + (bx0 >> f.hs[pl]);
      ch.w = (int)(gbw >> f.hs[pl]); ch.h = (int)(gbh >> f.vs[pl]); ch.stride = (int)f.bw;
      before_channel(ch, c);
      DecodeChannelCoop(br, state, T, mc, ch, c);
    }
  }
  // ---- ModularLfGroup (dec_frame.cc ProcessDCGroup: between the LF coefficients and the HF metadata): the sub-channels of the frame's extra channels that Squeeze
  // halved three times or more in both directions (cjxl squeezes a progressive or lossy alpha channel).  Nothing is in the stream when no channel falls in that range.
  if (f.mod_nchan > f.mod_global_decodable) {
    const uint32_t first_c = f.mod_global_decodable, px0 = bx0 * 8, py0 = by0 * 8;
    ChannelDesc d;
    uint32_t nch_lf = 0, widest = 0;
    for (uint32_t c = first_c; c < f.mod_nchan; c++) if (ModUnitRect(f, c, px0, py0, 2048, 3, 1000, &d)) { nch_lf++; widest = max(widest, (uint32_t)d.w); }
    if (nch_lf) {
      if (lane == 0) {
        if (has_lf && state != 0x130000u) { SetError(f, kErrAnsFinalState); s_fail = 1; }
        BitReader tmp;
        tmp.Init(f.cs, br.BitPos(), f.cs_size);
        if (!ReadGroupHeader(tmp, s_gh) || !s_gh.use_global_tree || s_gh.ntransforms != 0 || (with_refs && nch_lf > 4)) { SetError(f, kErrUnsupported); s_fail = 1; }
        br.Init(f.cs, tmp.BitPos(), sec_end);
        state = f.mod_code.use_prefix ? 0x130000u : br.Read(32);
      }
      WaveSync();
      if (s_fail) return;
      mc.wp = s_gh.wp; mc.stream_id = 1 + f.num_lf_groups + g;
      if (f.mod_code.lz77 && lane == 0) lz.Init(lz.window, widest);
      WaveSync();
      int k = 0;
      for (uint32_t c = first_c; c < f.mod_nchan; c++) if (ModUnitRect(f, c, px0, py0, 2048, 3, 1000, &d)) {
        before_channel(d, k);
        DecodeChannelCoop(br, state, T, mc, d, k);
        k++;
      }
    }
  }
  // ---- HF metadata: 4 channels {ytox, ytob, (strategy,hf_mul-1) x nb_blocks, sharpness}
  if (lane == 0) {
    if (state != 0x130000u) { SetError(f, kErrAnsFinalState); s_fail = 1; }
    s_u[1] = 1 + br.Read(CeilLog2D(gbw * gbh));
    BitReader tmp;
    tmp.Init(f.cs, br.BitPos(), f.cs_size);
    if (!ReadGroupHeader(tmp, s_gh) || !s_gh.use_global_tree || s_gh.ntransforms != 0) { SetError(f, kErrUnsupported); s_fail = 1; }
    br.Init(f.cs, tmp.BitPos(), sec_end);
    state = f.mod_code.use_prefix ? 0x130000u : br.Read(32);
  }
  WaveSync();
  if (s_fail) return;
  const uint32_t nb_blocks = s_u[1];
  const uint32_t mcw = (gbw + 7) / 8, mch = (gbh + 7) / 8;
  if (f.mod_code.lz77 && lane == 0) lz.Init(lz.window, max(max(nb_blocks, gbw), mcw));
  WaveSync();
  int32_t* m_ytox = scratch + 16;
  int32_t* m_ytob = m_ytox + mcw * mch;
  int32_t* m_blk = m_ytob + mcw * mch;
  int32_t* m_sharp = m_blk + 2 * nb_blocks;
  mc.wp = s_gh.wp; mc.stream_id = 1 + 2 * f.num_lf_groups + g;
  for (int k = 0; k < 4; k++) {       // (one call site: DecodeChannelCoop is inlined)
    ChannelDesc ch;
    ch.data = k == 0 ? m_ytox : k == 1 ? m_ytob : k == 2 ? m_blk : m_sharp;
    ch.w = k < 2 ? (int)mcw : k == 2 ? (int)nb_blocks : (int)gbw; ch.h = k < 2 ? (int)mch : k == 2 ? 2 : (int)gbh; ch.stride = ch.w;
    ch.hs = 0; ch.vs = 0;
    before_channel(ch, k);
    DecodeChannelCoop(br, state, T, mc, ch, k);
  }
  if (lane == 0) {
    if (state != 0x130000u) SetError(f, kErrAnsFinalState);
    else if (br.BitPos() > limit) SetError(f, kErrOverrun);
    else { if (f.single_section) f.stream_end_bitpos[0] = br.BitPos(); scratch[1] = (int32_t)nb_blocks; }
  }
  WaveSync();
}

// ---- varblock placement -----------------------------------------------------------------------------------------------------------------
// The varblocks of an LF group arrive as one list in raster order of their top-left blocks ("the next entry goes to the first block no
// earlier entry covers").  That is sequential — but varblocks never cross a 32-row band (they stay inside a 256x256-pixel group), so a
// band is complete before the scan enters the next one, and the list index at which a band starts is where the running sum of block areas
// reaches (rows above) x (group width): a prefix sum.  Three kernels:
//   LfBandStartKernel   one wavefront per LF group: wave-wide scans over the strategy list -> start index of each of its (up to 8) bands
//   LfPlaceSimtKernel   one band per LANE, 16 lanes per wavefront: every lane walks its band's scan (coverage bitmap of 32 rows x 256
//                       blocks = 1 KB of LDS per lane) and emits one 16-byte record per varblock; the walk is a serial chain per band, so
//                       like the entropy decode it runs in lock step on a few hundred wavefronts instead of one wavefront per chain
//   LfPlaceExpandKernel one workgroup per band: records -> coefficient offsets, per-group varblock lists (incl. the HF block-context
//                       bucket), block-info words of all covered blocks (sharpness merged), the band's rows of the chroma-from-luma maps
struct BandGeom { uint32_t gx, gy, bx0, by0, gbw, gbh, y_begin, y_end; };
__device__ __forceinline__ BandGeom BandGeometry(const FrameDev& f, uint32_t g, uint32_t band) {
  BandGeom b;
  b.gx = g % f.xlfgroups; b.gy = g / f.xlfgroups;
  b.bx0 = b.gx * 256; b.by0 = b.gy * 256;
  b.gbw = min(256u, f.bw - b.bx0); b.gbh = min(256u, f.bh - b.by0);
  b.y_begin = band * 32; b.y_end = min(b.gbh, b.y_begin + 32);
  return b;
}
// records of a band: a region of (rows x width) entries at (blocks of the LF groups above) + (of those to the left) + (of the bands above)
__device__ __forceinline__ size_t BandRecordBase(const FrameDev& f, const BandGeom& b) { return (size_t)b.by0 * f.bw + (size_t)b.bx0 * b.gbh + (size_t)b.y_begin * b.gbw; }
__device__ __forceinline__ uint32_t StrategyGeo(uint32_t s) {   // cx | cy << 8 | {log2 cx << 5 | log2 (cx cy) << 8 | order bucket << 12} << 16
  return CoveredX(s) | (CoveredY(s) << 8) | (((Log2CoveredX(s) << 5) | ((Log2CoveredX(s) + Log2CoveredY(s)) << 8) | (OrderBucket(s) << 12)) << 16);
}

__global__ __launch_bounds__(256) void LfBandStartKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular) return;
  const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (g >= f.num_lf_groups) return;
  const BandGeom bg = BandGeometry(f, g, 0);
  int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
  const uint32_t nb_blocks = (uint32_t)LdG(scratch + 1);
  uint32_t* starts = f.band_start + g * 8;
  if (LdG(f.status) != 0 || nb_blocks == 0 || nb_blocks > bg.gbw * bg.gbh) { if (lane < 8) StG(starts + lane, 0xFFFFFFFFu); return; }   // (a failed stream leaves no usable block count)
  const uint32_t mcw = (bg.gbw + 7) / 8, mch = (bg.gbh + 7) / 8;
  const int32_t* m_blk = scratch + 16 + 2 * mcw * mch;
  const uint32_t nbands = (bg.gbh + 31) / 32;
  if (lane == 0) StG(starts, 0u);
  uint32_t next = 1, cum = 0;                     // next band whose start is looked for; areas of the entries before `base`
  for (uint32_t base = 0; base < nb_blocks && next < nbands; base += 64) {
    const uint32_t idx = base + lane;
    uint32_t area = 0;
    if (idx < nb_blocks) { const uint32_t st = (uint32_t)LdG(m_blk + idx); if (st < 27) area = CoveredX(st) * CoveredY(st); }
    uint32_t incl = area;                         // inclusive prefix sum over the wavefront
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) { const uint32_t o = __shfl_up(incl, d, 64); if ((int)lane >= d) incl += o; }
    const uint32_t total = __shfl(incl, 63, 64);
    while (next < nbands) {                       // (several bands may start inside one chunk of 64 entries)
      const uint32_t target = next * 32 * bg.gbw;
      if (cum + total < target) break;
      const uint64_t reach = __ballot(cum + incl - area >= target && idx < nb_blocks);   // entries that start at or past the band's first block
      if (!reach) break;                          // the band starts exactly where this chunk ends (or the list does)
      const int l0 = __ffsll((long long)reach) - 1;
      const uint32_t at = __shfl(cum + incl - area, l0, 64);
      if (lane == 0) StG(starts + next, at == target ? base + (uint32_t)l0 : 0xFFFFFFFFu);   // not exactly there: a varblock straddles the band's top (damaged list)
      next++;
    }
    cum += total;
  }
  if (lane == 0) for (uint32_t k = next; k < 8; k++) StG(starts + k, k < nbands ? 0xFFFFFFFFu : 0u);     // bands the list never reaches
}

constexpr uint32_t kPlaceLanes = 16, kPlaceLaneWords = 256 + 17;          // per lane: 32 rows x 8 words of coverage, 8 group offsets, 8 group counts (odd stride: no bank conflicts)
__global__ __launch_bounds__(64) void LfPlaceSimtKernel(const FrameDev* __restrict__ frames, const uint2* __restrict__ units, uint32_t num_units) {
  __shared__ uint32_t s_lds[kPlaceLanes * kPlaceLaneWords];
  const uint32_t u = blockIdx.x * kPlaceLanes + threadIdx.x;
  if (threadIdx.x >= kPlaceLanes || u >= num_units) return;
  const uint2 unit = LdG(units + u);
  const FrameDev& f = frames[unit.x];
  const uint32_t g = unit.y & 0xFFFF, band = unit.y >> 16;
  const BandGeom bg = BandGeometry(f, g, band);
  uint32_t* cnt_out = f.place_cnt + g * 8 + band;
  uint32_t k = LdG(f.band_start + g * 8 + band);
  if (k == 0xFFFFFFFFu) { if (LdG(f.status) == 0) SetError(f, kErrVarblock); StG(cnt_out, 0u); return; }
  int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
  const uint32_t nb_blocks = (uint32_t)LdG(scratch + 1);
  const uint32_t mcw = (bg.gbw + 7) / 8, mch = (bg.gbh + 7) / 8;
  const int32_t* m_blk = scratch + 16 + 2 * mcw * mch;
  uint4* rec = f.place_rec + BandRecordBase(f, bg);
  uint32_t* ring = s_lds + threadIdx.x * kPlaceLaneWords;
  uint32_t* goff = ring + 256;
  uint32_t* gcnt = ring + 264;
  for (uint32_t i = 0; i < 256; i++) { const uint32_t wi = i & 7; ring[i] = wi * 32 >= bg.gbw ? ~0u : (bg.gbw - wi * 32 < 32 ? ~0u << (bg.gbw - wi * 32) : 0u); }   // bits outside the group are pre-set
  for (uint32_t i = 0; i < 8; i++) { goff[i] = 0; gcnt[i] = 0; }
  uint32_t y = bg.y_begin, wi = 0, count = 0, flags_acc = 0, err = 0;
  int2 sq_next = k < nb_blocks ? make_int2(LdG(m_blk + k), LdG(m_blk + nb_blocks + k)) : make_int2(-1, -1);
  while (y < bg.y_end) {
    const uint32_t row = (y & 31) * 8;
    const uint32_t cov = ring[row + wi];
    if (cov == ~0u) { if (++wi == 8) { wi = 0; y++; } continue; }     // the first uncovered block of a row only moves right
    const uint32_t xb = (uint32_t)__ffs((int)~cov) - 1, x = wi * 32 + xb;
    if (k >= nb_blocks) { err = kErrVarblock; break; }
    const uint32_t s = (uint32_t)sq_next.x, q = (uint32_t)sq_next.y;
    k++;
    if (k < nb_blocks) sq_next = make_int2(LdG(m_blk + k), LdG(m_blk + nb_blocks + k));     // (arrives while this varblock is placed)
    if (s >= 27 || q > 255) { err = kErrBadValue; break; }           // (negative values wrap to large ones)
    if (f.subsampled && s != 0) { err = kErrUnsupported; break; }    // chroma-subsampled frames: 8x8 DCT only
    const uint32_t geo = StrategyGeo(s), cx = geo & 0xFF, cy = (geo >> 8) & 0xFF;
    if (x + cx > bg.gbw || y + cy > bg.gbh || xb + cx > 32 || (y % 32) + cy > 32) { err = kErrVarblock; break; }
    const uint32_t bits = (cx == 32 ? ~0u : (1u << cx) - 1u) << xb;
    uint32_t clash = 0;
    for (uint32_t iy = 0; iy < cy; iy++) { const uint32_t o = ((y + iy) & 31) * 8 + wi; const uint32_t wv = ring[o]; clash |= wv & bits; ring[o] = wv | bits; }
    if (clash) { err = kErrVarblock; break; }
    if ((x % 4) + cx > 4 || (y % 4) + cy > 4) flags_acc |= 4u;      // not inside one 32x32 tile: the IDCT uses 64x64 tiles
    if (s == 1 || s == 2 || s == 3 || (s >= 12 && s <= 17)) flags_acc |= 8u;  // IDENTITY / DCT2X2 / DCT4X4 / DCT4X8 / DCT8X4 / AFV0-3: the tile kernel variant that carries them
    if (s == 1 || s == 2 || (s >= 14 && s <= 17)) flags_acc |= 16u;           // IDENTITY / DCT2X2 / AFV (statistics; until round 5 a second kernel redid these blocks after the tile kernel)
    if (cx > 8 || cy > 8) flags_acc |= (x % 8) || (y % 8) ? 3u : 2u;   // DCT128/256 family: BigIdctKernel (+ generic path if unaligned)
    else if ((x % 8) + cx > 8 || (y % 8) + cy > 8) flags_acc |= 1u;  // varblock not contained in a 64x64 tile: generic IDCT
    const uint32_t go = goff[wi], gc = gcnt[wi];
    goff[wi] = go + cx * cy * 64;
    gcnt[wi] = gc + 1;
    // record: {x | y << 8 | s << 16 | q << 24, coefficient offset, index in the group's list, geometry word}
    StG(rec + count, make_uint4(x | (y << 8) | (s << 16) | (q << 24), go, gc, geo));
    count++;
  }
  if (err) { SetError(f, err); count = 0; }
  if (flags_acc) atomicOr(f.frame_flags, flags_acc);
  StG(cnt_out, count);
  for (uint32_t i = 0; i < 8; i++) if (i * 32 < bg.gbw) StG(f.vb_count + (bg.gy * 8 + band) * f.xgroups + bg.gx * 8 + i, err ? 0u : gcnt[i]);
}

// The same walk with one band per WAVEFRONT (round 6, latency mode: a handful of frames).  The lane-per-band form above waits for a global load per varblock (the next list entry
// is requested one step ahead; the walk of a band of a 4K frame — up to 8192 blocks — took 7 ms, a tenth of a single image's decode).  Here every lane runs the walk (scalar
// registers), the (strategy, quantiser) list sits in two VGPRs 64 entries at a time, the coverage ring in LDS, the running offsets / counts of the band's eight groups in two VGPRs
// (lane = group column).  Records, counts, flags and errors exactly as LfPlaceSimtKernel.
__global__ __launch_bounds__(64) void LfPlaceWaveKernel(const FrameDev* __restrict__ frames, const uint2* __restrict__ units, uint32_t num_units) {
  __shared__ uint32_t ring[256];
  const uint32_t u = blockIdx.x, lane = threadIdx.x;
  if (u >= num_units) return;
  const uint2 unit = LdG(units + u);
  const FrameDev& f = frames[unit.x];
  const uint32_t g = unit.y & 0xFFFF, band = unit.y >> 16;
  const BandGeom bg = BandGeometry(f, g, band);
  uint32_t* cnt_out = f.place_cnt + g * 8 + band;
  uint32_t k = Uniform(LdG(f.band_start + g * 8 + band));
  if (k == 0xFFFFFFFFu) { if (lane == 0) { if (LdG(f.status) == 0) SetError(f, kErrVarblock); StG(cnt_out, 0u); } return; }
  int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
  const uint32_t nb_blocks = Uniform((uint32_t)LdG(scratch + 1));
  const uint32_t gbw = Uniform(bg.gbw), gbh = Uniform(bg.gbh), y_end = Uniform(bg.y_end);
  const uint32_t mcw = (gbw + 7) / 8, mch = (gbh + 7) / 8;
  const int32_t* m_blk = scratch + 16 + 2 * mcw * mch;
  uint4* rec = f.place_rec + BandRecordBase(f, bg);
  for (uint32_t i = lane; i < 256; i += 64) { const uint32_t wi = i & 7; ring[i] = wi * 32 >= gbw ? ~0u : (gbw - wi * 32 < 32 ? ~0u << (gbw - wi * 32) : 0u); }   // bits outside the group are pre-set
  __syncthreads();
  uint32_t goffv = 0, gcntv = 0;          // lane = group column of the band
  uint32_t y = Uniform(bg.y_begin), wi = 0, count = 0, flags_acc = 0, err = 0;
  uint32_t kbase = k & ~63u;
  int32_t sv = kbase + lane < nb_blocks ? LdG(m_blk + kbase + lane) : -1, qv = kbase + lane < nb_blocks ? LdG(m_blk + nb_blocks + kbase + lane) : -1;
  __builtin_amdgcn_s_waitcnt(0x0F70);
  const bool subsampled = f.subsampled != 0;
  while (y < y_end) {
    const uint32_t row = (y & 31) * 8;
    const uint32_t cov = Uniform(ring[row + wi]);
    if (cov == ~0u) { if (++wi == 8) { wi = 0; y++; } continue; }     // the first uncovered block of a row only moves right
    const uint32_t xb = (uint32_t)__ffs((int)~cov) - 1, x = wi * 32 + xb;
    if (k >= nb_blocks) { err = kErrVarblock; break; }
    if (k >= kbase + 64) {
      kbase = k & ~63u;
      sv = kbase + lane < nb_blocks ? LdG(m_blk + kbase + lane) : -1; qv = kbase + lane < nb_blocks ? LdG(m_blk + nb_blocks + kbase + lane) : -1;
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const uint32_t s = (uint32_t)__builtin_amdgcn_readlane(sv, (int)(k - kbase)), q = (uint32_t)__builtin_amdgcn_readlane(qv, (int)(k - kbase));
    k++;
    if (s >= 27 || q > 255) { err = kErrBadValue; break; }           // (negative values wrap to large ones)
    if (subsampled && s != 0) { err = kErrUnsupported; break; }      // chroma-subsampled frames: 8x8 DCT only
    const uint32_t geo = StrategyGeo(s), cx = geo & 0xFF, cy = (geo >> 8) & 0xFF;
    if (x + cx > gbw || y + cy > gbh || xb + cx > 32 || (y % 32) + cy > 32) { err = kErrVarblock; break; }
    const uint32_t bits = (cx == 32 ? ~0u : (1u << cx) - 1u) << xb;
    uint32_t clash = 0;
    for (uint32_t iy = 0; iy < cy; iy++) { const uint32_t o = ((y + iy) & 31) * 8 + wi; const uint32_t wv = Uniform(ring[o]); clash |= wv & bits; if (lane == 0) ring[o] = wv | bits; }
    if (clash) { err = kErrVarblock; break; }
    if ((x % 4) + cx > 4 || (y % 4) + cy > 4) flags_acc |= 4u;
    if (s == 1 || s == 2 || s == 3 || (s >= 12 && s <= 17)) flags_acc |= 8u;
    if (s == 1 || s == 2 || (s >= 14 && s <= 17)) flags_acc |= 16u;
    if (cx > 8 || cy > 8) flags_acc |= (x % 8) || (y % 8) ? 3u : 2u;
    else if ((x % 8) + cx > 8 || (y % 8) + cy > 8) flags_acc |= 1u;
    const uint32_t go = (uint32_t)__builtin_amdgcn_readlane((int)goffv, (int)wi), gc = (uint32_t)__builtin_amdgcn_readlane((int)gcntv, (int)wi);
    goffv = lane == wi ? go + cx * cy * 64 : goffv;
    gcntv = lane == wi ? gc + 1 : gcntv;
    if (lane == 0) StG(rec + count, make_uint4(x | (y << 8) | (s << 16) | (q << 24), go, gc, geo));
    count++;
  }
  if (lane == 0) {
    if (err) { SetError(f, err); count = 0; }
    if (flags_acc) atomicOr(f.frame_flags, flags_acc);
    StG(cnt_out, count);
  }
  if (lane < 8 && lane * 32 < gbw) StG(f.vb_count + (bg.gy * 8 + band) * f.xgroups + bg.gx * 8 + lane, err ? 0u : gcntv);
}

__global__ __launch_bounds__(256) void LfPlaceExpandKernel(const FrameDev* __restrict__ frames, const uint2* __restrict__ units, uint32_t num_units) {
  const uint2 unit = LdG(units + blockIdx.x);
  const FrameDev& f = frames[unit.x];
  const uint32_t g = unit.y & 0xFFFF, band = unit.y >> 16;
  const BandGeom bg = BandGeometry(f, g, band);
  int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
  const uint32_t nb_blocks = (uint32_t)LdG(scratch + 1);
  if (LdG(f.status) != 0 || nb_blocks == 0 || nb_blocks > bg.gbw * bg.gbh) return;
  const uint32_t mcw = (bg.gbw + 7) / 8, mch = (bg.gbh + 7) / 8;
  const int32_t* m_ytox = scratch + 16;
  const int32_t* m_ytob = m_ytox + mcw * mch;
  const int32_t* m_sharp = m_ytob + mcw * mch + 2 * nb_blocks;
  // ---- chroma-from-luma maps: the band's four tile rows
  for (uint32_t i = band * 4 * mcw + threadIdx.x; i < min(mch, band * 4 + 4) * mcw; i += blockDim.x) {
    const uint32_t y = i / mcw, x = i % mcw;
    const int a = m_ytox[i], b = m_ytob[i];
    if (a < -128 || a > 127 || b < -128 || b > 127) SetError(f, kErrBadValue);
    if (f.subsampled && (a | b)) SetError(f, kErrUnsupported);      // chroma from luma across different block grids
    const size_t o = (size_t)(bg.gy * 32 + y) * f.cw + bg.gx * 32 + x;
    f.ytox[o] = (int8_t)a; f.ytob[o] = (int8_t)b;
  }
  const uint32_t count = LdG(f.place_cnt + g * 8 + band);
  const uint4* rec = f.place_rec + BandRecordBase(f, bg);
  const BlockCtxDev& bcm = *f.bcm;
  for (uint32_t i = threadIdx.x; i < count; i += blockDim.x) {
    const uint4 r = LdG(rec + i);
    const uint32_t x = r.x & 0xFF, y = (r.x >> 8) & 0xFF, s = (r.x >> 16) & 0xFF, q = r.x >> 24;
    const uint32_t cx = r.w & 0xFF, cy = (r.w >> 8) & 0xFF;
    const size_t o = (size_t)(bg.by0 + y) * f.bw + bg.bx0 + x;
    // block-context inputs of the HF stage (ac_context.h): quant-field and LF-value buckets of the varblock's first block,
    // folded into qf_idx * num_lf_ctxs + lf_idx (< 64) here so that the HF decoder's loop has no threshold searches
    uint32_t qf_idx = 0;
    for (uint32_t t = 0; t < bcm.n_qf_thr; t++) qf_idx += q + 1 > bcm.qf_thr[t];
    uint32_t lf_idx = 0;
    if (bcm.num_lf_ctxs > 1 && !f.use_lf_frame) {      // (dec_cache.cc: quant_dc is zero-filled when the LF image is an LF frame's)
      auto lfq_at = [&](int c) { return LdG(f.lfq[c] + (size_t)((bg.by0 + y) >> f.vs[c]) * f.bw + ((bg.bx0 + x) >> f.hs[c])); };   // (quant_dc is kept at full resolution)
      const int32_t q0 = lfq_at(0), q1 = lfq_at(1), q2 = lfq_at(2);
      uint32_t b0 = 0, b1 = 0, b2 = 0;
      for (uint32_t t = 0; t < bcm.n_lf_thr[0]; t++) b0 += q0 > bcm.lf_thr[0][t];
      for (uint32_t t = 0; t < bcm.n_lf_thr[1]; t++) b1 += q1 > bcm.lf_thr[1][t];
      for (uint32_t t = 0; t < bcm.n_lf_thr[2]; t++) b2 += q2 > bcm.lf_thr[2][t];
      lf_idx = (b0 * (bcm.n_lf_thr[2] + 1) + b2) * (bcm.n_lf_thr[1] + 1) + b1;
    }
    const uint32_t qlf = (qf_idx * bcm.num_lf_ctxs + lf_idx) & 63u;
    // per-group varblock list for the HF decoder (everything its block start needs, ready to unpack):
    // {strategy | log2 cx << 5 | log2 (cx cy) << 8 | order bucket << 12 | x << 16 | y << 21 | qlf << 26, coefficient offset}
    const uint32_t gg = (bg.gy * 8 + y / 32) * f.xgroups + bg.gx * 8 + x / 32;
    StG(f.vb_list + (size_t)gg * 1024 + r.z, make_uint2(s | (r.w >> 16) | ((x % 32) << 16) | ((y % 32) << 21) | (qlf << 26), r.y));
    for (uint32_t iy = 0; iy < cy; iy++) for (uint32_t ix = 0; ix < cx; ix++) {
      // (the sharpness map is merged here: every covered block's word is written exactly once)
      const int32_t sh = LdG(m_sharp + (size_t)(y + iy) * bg.gbw + x + ix);
      if (sh < 0 || sh > 7) SetError(f, kErrBadValue);
      StG(f.blk_info + o + (size_t)iy * f.bw + ix, PackBlockInfo(s, ix == 0 && iy == 0, q, ix, iy, (uint32_t)sh & 7u));
      StG(f.coef_off + o + (size_t)iy * f.bw + ix, r.y);   // (every covered block: the IDCT reads info and offset side by side)
    }
  }
}

// Two wavefronts per workgroup share four consecutive LF groups: wavefront w decodes groups w and 3 - w one after the
// other.  A 3840x2160 frame has two large LF groups (256x256 and 224x256 blocks) and two slivers (14 rows): with one
// wavefront per group the sliver wavefronts were done after 5 % of the kernel while their LDS stayed allocated; paired
// like this both wavefronts carry about the same load and a frame holds 36 KB of LDS instead of 52 KB, so that the LF
// workgroups of two batches in flight (2 x 36 KB) and an HF workgroup (80 KB) fit one CU.  (The kernel needs 272 VGPRs,
// one wavefront per SIMD: a CU never hosts more than two of these workgroups, whatever the dispatcher would like.)
// Small launches (single images) take one group per wavefront instead: latency over LDS economy.
// CAPPED: at most 170 VGPRs (with spills) so that pixel-kernel wavefronts of the batch on the main stream fit the same SIMDs — the
// variant for large pipelined batches; single images take the uncapped one (267 VGPRs, LF stage 20 % shorter).
template <bool CAPPED> __global__ __launch_bounds__(64 * kLfDecWaves, CAPPED ? JXL_LF_MINW : 1) void LfDecodeKernel(const FrameDev* __restrict__ frames, uint32_t groups_per_block, uint32_t tree_cap, uint32_t lds_bytes, int take_simt_frames, uint32_t wp_base) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || (f.lf_simt && !take_simt_frames)) return;
  const uint32_t first = blockIdx.x * groups_per_block;
  if (first >= f.num_lf_groups) return;
  // take_simt_frames 2: of the SIMT kernel's frames only the streams it handed back (lf_scratch[2] == kLfRedoMark: values outside the range of its
  // 32-bit weighted-predictor arithmetic, explicit predictor parameters) — normally none, and the workgroup ends before it stages anything
  const bool redo_only = f.lf_simt && take_simt_frames == 2;
  if (redo_only) {
    bool any = false;
    for (uint32_t k = 0; k < groups_per_block && first + k < f.num_lf_groups; k++) any |= LdG(f.lf_scratch + (uint64_t)(first + k) * f.lf_scratch_stride + 2) == kLfRedoMark;
    if (!any) return;
  }
  ModTables T;
  StageModular(f.tree, f.tree_nodes, f.mod_code, T, tree_cap, lds_bytes);
  if (wp_base) T.wp_off = wp_base + (threadIdx.x >> 6) * kWpLdsBytes;      // weighted-predictor rows of this wavefront's stream (batches with such trees)
  __shared__ int s_fail_w[kLfDecWaves];
  __shared__ GroupHeaderD s_gh_w[kLfDecWaves];
  __shared__ uint32_t s_u_w[kLfDecWaves][4];
  const uint32_t wave = threadIdx.x >> 6;
  for (uint32_t turn = 0; turn < groups_per_block / kLfDecWaves; turn++) {   // (no block-wide barrier after this point)
    const uint32_t local = turn & 1 ? groups_per_block - 1 - (turn / 2) * kLfDecWaves - wave : (turn / 2) * kLfDecWaves + wave;
    if (first + local >= f.num_lf_groups) continue;
    if (redo_only && LdG(f.lf_scratch + (uint64_t)(first + local) * f.lf_scratch_stride + 2) != kLfRedoMark) continue;
    LfDecodeGroup(f, first + local, T, s_fail_w[wave], s_gh_w[wave], s_u_w[wave]);
  }
}

// =====================================================================================================================
// K_lf, SIMT form: one LF-group stream per LANE (LfSimtPlan, kernels.h).  An LF group of a 4K frame is one rANS stream of ~275 000
// tokens whose every context depends on the sample before it: one wavefront per stream (LfDecodeKernel above) keeps 1024 wavefronts
// with one busy lane each, 170 VGPRs and 18 KB of LDS apiece, on the chip for the whole stage.  Here a wavefront carries up to 64
// streams in lock step — one token per lane per iteration — so that a batch of 256 frames needs a few dozen wavefronts, no LDS and
// ~50 VGPRs: the stage's latency grows (every iteration pays two dependent L2 round trips: property -> cluster table, alias table),
// its footprint shrinks by more than an order of magnitude, and the stages of other batches get the CUs.
// Per lane: the bit reader (next word always in flight), the rANS state, the position in the channel, W / NW and three prefetched
// samples of the row above (loaded three iterations ahead, across the row boundary), pointers to the tables of its frame.
// Everything else (geometry, sub-stream headers, channel classes) is recomputed from the descriptors in the rare path that
// runs when a lane reaches the end of a row.  Semantics = DecodeChannelCoop's fast path (= jxl_dev.h DecodeModularChannel).
// =====================================================================================================================
// Bit reader of a SIMT lane: as BitReaderP (next word always in flight), but the refill load is unconditional — the word index is
// clamped to the end of the CODESTREAM, which is followed by 80 zero bytes (Codestream::storage) — so that the loaded word lands in
// `nextw` without a select (a select would make every refill wait for its own load).  A reader that runs past its section reads the
// bytes of the next one instead of zeros; that only happens on damaged streams, which the end-of-stream position check rejects.
struct BitReaderQ {
  const uint32_t* words;
  uint32_t wpos, wlast, nextw;
  uint64_t buf;
  int avail;
  __device__ __forceinline__ uint32_t Load(uint32_t i) const { return LdG(words + min(i, wlast)); }
  __device__ __forceinline__ void Init(const uint8_t* base, uint64_t bit_pos, uint64_t cs_size) {
    words = reinterpret_cast<const uint32_t*>(base);
    wpos = (uint32_t)(bit_pos >> 5);
    wlast = (uint32_t)((cs_size + 3) >> 2);        // first word of the zero padding
    buf = (uint64_t)Load(wpos) | ((uint64_t)Load(wpos + 1) << 32);
    wpos += 2;
    nextw = Load(wpos);
    avail = 64;
    const int skip = (int)(bit_pos & 31);
    buf >>= skip; avail -= skip;
    Refill();
  }
  __device__ __forceinline__ void Refill() {
    if (avail <= 32) {
      buf |= (uint64_t)nextw << avail;
      avail += 32;
      wpos++;
      nextw = Load(wpos);
    }
  }
  __device__ __forceinline__ uint32_t Read(int n) {  // n <= 32
    Refill();
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; avail -= n;
    return v;
  }
  __device__ __forceinline__ uint64_t BitPos() const { return (uint64_t)wpos * 32 - (uint64_t)avail; }
};
__device__ __forceinline__ bool SkipGroupHeaderSimt(BitReaderQ& br, bool* default_wp) {   // GroupHeader of an eligible stream: global tree, no transforms
  const uint32_t use_global_tree = br.Read(1);
  *default_wp = br.Read(1) != 0;
  if (!*default_wp) { br.Read(10); br.Read(25); br.Read(16); }         // explicit weighted-predictor parameters (streams that use the predictor go back to LfDecodeKernel)
  const uint32_t sel = br.Read(2);
  const uint32_t ntr = sel == 0 ? 0u : sel == 1 ? 1u : sel == 2 ? 2u + br.Read(4) : 18u + br.Read(8);
  return use_global_tree && ntr == 0;
}

// ---- weighted predictor of a SIMT lane (context_predict.h weighted::State with the header's default parameters — p1 16, p2 10, p3 7 7 7 0 0,
// weights 13 12 12 12; a stream whose group header carries other parameters is handed back to LfDecodeKernel).  The reference keeps, per
// sub-predictor, error magnitudes of the row above and of this row: a sample's magnitude is stored at its own position and ADDED to the
// position right of it in the row above, i.e. to what the next sample reads as "N".  Per lane that is a register chain — aN (stored value
// + the left neighbour's magnitude), aNW (the aN of the sample before) — and the stored values of the row above arrive from the stream's
// two rows of 32-byte records {four magnitudes, true error} in global memory, two positions ahead of their use; the sample's own record
// goes out with one 16-byte and one 4-byte store.  32-bit arithmetic throughout, exact while |sample| <= 2^20 and |true error| <= 2^24
// (products with the 5-bit weights stay below 2^31); the lane checks both per sample and hands the stream back otherwise.
constexpr uint32_t kWpRecBytes = 32, kWpRowBytes = 258 * kWpRecBytes;
struct WpLane {
  uint32_t aN[4], aNW[4], mA[4], mB[4];     // mA: stored magnitudes of the row above at x + 1 (arrived), mB: at x + 2 (in flight)
  int32_t teN, teNW, teW, eA, eB;           // true errors at N, NW, W; of the row above at x + 1 / x + 2
  int32_t pr[4], avg;                       // this sample's sub-predictions and their weighted average (x 8)
};

// One LF-group stream per lane, `lanes_per_wave` lanes per wavefront.  Instantiations: <false, false> the lean one — a context per row or a table
// over W + N - NW, predictors zero / W / clamped gradient (the gradient trees of `cjxl --faster_decoding`, the synthesiser's default): 72 VGPRs (under a 64-register cap the row-change path spilled six);
// <false, true> adds tables over W / N and pairs of properties and the rarer predictors; <true, true> weighted-predictor state on top (any channel
// class with kLfSimtWpLive): a 256-byte division table and ~50 VGPRs more.
// No LDS in any of them (kernels.h: LF workgroups stay for hundreds of milliseconds and would fragment what the HF workgroups need): the
// weighted predictor's 64-entry division table is read from global memory (g_wp_div: it stays in the vector L1, and its reads hang off
// the prediction, which only meets the entropy chain when the token is there).
// QUAD (weighted-predictor launches): four adjacent lanes carry ONE stream.  Everything but the weighted predictor is computed redundantly by the four (a wavefront
// instruction costs the same for one or four active lanes); of the predictor, lane q owns sub-predictor q — its error magnitudes, weight, prediction — and the sums over
// the four (weights, weighted predictions) are two DPP quad-permute additions each.  ~100 of the ~330 instructions per sample go away; 8 streams occupy 32 lanes.
__device__ __forceinline__ int32_t QuadSum(int32_t v) {
  v += __builtin_amdgcn_mov_dpp(v, 0xB1, 0xF, 0xF, true);      // quad_perm [1, 0, 3, 2]
  v += __builtin_amdgcn_mov_dpp(v, 0x4E, 0xF, 0xF, true);      // quad_perm [2, 3, 0, 1]
  return v;
}
template <bool WP, bool GEN, bool QUAD = false> __global__ __launch_bounds__(256, WP ? 4 : (GEN ? 6 : 7)) void LfDecodeSimtKernel(const FrameDev* __restrict__ frames, const LfSimtStream* __restrict__ streams, const LfSimtLane* __restrict__ lanes,
                                                          const uint8_t* __restrict__ luts, uint32_t num_lanes, uint32_t lanes_per_wave, int high_priority, const uint32_t* __restrict__ s_div) {
  static_assert(!QUAD || WP, "quads only pay for the weighted predictor");
  const uint32_t lane_in_wave = QUAD ? (threadIdx.x & 63) >> 2 : (threadIdx.x & 63);
  const uint32_t q = QUAD ? threadIdx.x & 3 : 0;              // which sub-predictor this lane owns
  const uint32_t li = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * lanes_per_wave + lane_in_wave;
  if (lane_in_wave >= lanes_per_wave || li >= num_lanes) return;
  if (high_priority & 1) __builtin_amdgcn_s_setprio(3);
  const int xp = high_priority;     // experiments (JXL_HIP_LF_PRIO bits 2, 4: drop the sample stores / the row-above prefetch — wrong pixels, timing only)
  uint32_t cur, end;
  { const uint2 ln = LdG(reinterpret_cast<const uint2*>(lanes + li)); cur = ln.x; end = ln.x + ln.y; }
  BitReaderQ br;
  br.words = nullptr; br.wpos = br.wlast = br.nextw = 0; br.buf = 0; br.avail = 0;
  uint32_t state = 0;
  int32_t* out = nullptr;                 // current row of the current channel
  uint32_t x = 0, w = 0, y = 0, h = 0;
  int32_t stride = 0;
  int32_t left = 0, nw = 0, up0 = 0, up1 = 0, up2 = 0;
  uint32_t lut_off = 0;                   // table of the current row's class (byte offset in the blob)
  const uint64_t* alias = nullptr;
  const uint32_t* cfgp = nullptr;
  uint32_t cfgu = 0, la = 0, cls = 0, const_cluster = 0;     // cls: class word of the current row (kernels.h)
  bool hp = false, roll_next = false;     // a row above exists; the prefetch may run on into the next row (it exists and the row is >= 4 wide)
  int c = 7;                              // channel of the stream; 7: start the next stream
  WpLane wp;
  uint8_t* wrows = nullptr;               // this stream's two rows of weighted-predictor records
  uint8_t* wprev = nullptr; uint8_t* wcur = nullptr;
  if (WP) {
#pragma unroll
    for (int i = 0; i < 4; i++) wp.aN[i] = wp.aNW[i] = wp.mA[i] = wp.mB[i] = 0;
    wp.teN = wp.teNW = wp.teW = wp.eA = wp.eB = 0; wp.avg = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) wp.pr[i] = 0;
  }
  for (;;) {
    if (__builtin_expect(x >= w, 0)) {
      // ---- rare path: next row, channel, sub-stream or stream
      bool more = true;
      bool new_channel = false;
      for (;;) {
        if (c < 7 && y + 1 < h) { y++; out += stride; break; }
        const LfSimtStream* st = streams + cur;
        if (c == 6) {   // the stream's last channel is complete
          const FrameDev& f = frames[LdG(&st->frame)];
          const uint32_t g = LdG(&st->group);
          const uint64_t limit = (f.single_section ? f.cs_size : f.sec_off[1 + g] + f.sec_size[1 + g]) * 8;
          if (state != 0x130000u) SetError(f, kErrAnsFinalState);
          else if (br.BitPos() > limit) SetError(f, kErrOverrun);
          else if (f.single_section) f.stream_end_bitpos[0] = br.BitPos();
          cur++; st++; c = 7;
        }
        if (c >= 7) {
          if (cur >= end) { more = false; break; }
          const FrameDev& f = frames[LdG(&st->frame)];
          const uint32_t g = LdG(&st->group);
          br.Init(f.cs, f.single_section ? (f.mod_nchan ? f.stream_end_bitpos[1] : f.lf_start_bitpos) : f.sec_off[1 + g] * 8, f.cs_size);
          int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
          scratch[0] = (int32_t)br.Read(2);      // extra_precision
          scratch[1] = 0; scratch[2] = 0;
          bool default_wp = true;
          if (!SkipGroupHeaderSimt(br, &default_wp)) { SetError(f, kErrUnsupported); cur++; continue; }   // (c stays 7: the next stream)
          if (WP && !default_wp) { scratch[2] = kLfRedoMark; cur++; continue; }
          state = br.Read(32);
          alias = f.mod_code.alias; cfgp = f.mod_code.cfg; la = f.mod_code.log_alpha; cfgu = f.mod_cfg_uniform;
          if (WP) wrows = reinterpret_cast<uint8_t*>(f.wp_scratch + (uint64_t)g * f.wp_scratch_stride);
          c = -1;
        }
        c++;
        new_channel = true;
        const FrameDev& f = frames[LdG(&st->frame)];
        const uint32_t g = LdG(&st->group);
        const uint32_t gx = g % f.xlfgroups, gy = g / f.xlfgroups;
        const uint32_t bx0 = gx * 256, by0 = gy * 256;
        const uint32_t gbw = min(256u, f.bw - bx0), gbh = min(256u, f.bh - by0);
        int32_t* scratch = f.lf_scratch + (uint64_t)g * f.lf_scratch_stride;
        if (c == 3) {   // HF metadata sub-stream
          bool ok = true;
          if (state != 0x130000u) { SetError(f, kErrAnsFinalState); ok = false; }
          const uint32_t nb = 1 + br.Read(CeilLog2D(gbw * gbh));
          bool default_wp = true;
          if (ok && !SkipGroupHeaderSimt(br, &default_wp)) { SetError(f, kErrUnsupported); ok = false; }
          if (WP && ok && !default_wp) { scratch[2] = kLfRedoMark; ok = false; }
          if (!ok) { cur++; c = 7; continue; }
          scratch[1] = (int32_t)nb;
          state = br.Read(32);
        }
        uint32_t nw_ = 0, nh_ = 0;
        int32_t* data;
        if (c < 3) {
          const int pl = c == 0 ? 1 : c == 1 ? 0 : 2;          // stream channel order is Y, X, B
          data = f.lfq[pl] + (size_t)(by0 >> f.vs[pl]) * f.bw + (bx0 >> f.hs[pl]);
          nw_ = gbw >> f.hs[pl]; nh_ = gbh >> f.vs[pl]; stride = (int32_t)f.bw;
        } else {
          const uint32_t mcw = (gbw + 7) / 8, mch = (gbh + 7) / 8, nb = (uint32_t)scratch[1];
          int32_t* m_ytox = scratch + 16;
          if (c == 3) { data = m_ytox; nw_ = mcw; nh_ = mch; }
          else if (c == 4) { data = m_ytox + mcw * mch; nw_ = mcw; nh_ = mch; }
          else if (c == 5) { data = m_ytox + 2 * mcw * mch; nw_ = nb; nh_ = 2; }
          else { data = m_ytox + 2 * mcw * mch + 2 * nb; nw_ = gbw; nh_ = gbh; }
          stride = (int32_t)nw_;
        }
        y = 0; out = data;
        if (nw_ == 0 || nh_ == 0) { w = 0; h = 0; continue; }      // empty channel: on to the next one
        w = nw_; h = nh_;
        break;
      }
      if (!more) break;
      x = 0;
      hp = y > 0;
      roll_next = y + 1 < h && w >= 4;
      {  // class of this row
        uint2 e = LdG(reinterpret_cast<const uint2*>(&streams[cur].chan[c]));
        if (e.y & kLfSimtRows) {
          const uint32_t k = LdG(luts + e.x + min(y, 511u));
          e = LdG(reinterpret_cast<const uint2*>(luts + e.x + 512 + 8 * k));
        }
        lut_off = e.x; cls = e.y; const_cluster = (e.y >> 16) & 0xFF;
      }
      if (hp && w < 4) {                 // narrow rows: the rolling prefetch could run ahead of the stores
        up0 = LdG(out - stride); up1 = w > 1 ? LdG(out - stride + 1) : 0; up2 = w > 2 ? LdG(out - stride + 2) : 0;
      }
      if (WP && (cls & kLfSimtWpLive)) {
        if (w > 256) {                   // (the host keeps such channels off this kernel; a damaged stream may still announce one)
          const LfSimtStream* st = streams + cur;
          const FrameDev& f = frames[LdG(&st->frame)];
          (f.lf_scratch + (uint64_t)LdG(&st->group) * f.lf_scratch_stride)[2] = kLfRedoMark;
          cur++; c = 7; w = 0; h = 0; x = 0; continue;
        }
        if (new_channel) {               // fresh state: the row "above" row 0 holds zeros
          for (uint32_t k = 0; k < w; k++) { StG(reinterpret_cast<uint4*>(wrows + kWpRowBytes + k * kWpRecBytes), make_uint4(0, 0, 0, 0)); StG(reinterpret_cast<int32_t*>(wrows + kWpRowBytes + k * kWpRecBytes + 16), 0); }
        }
        wcur = wrows + (y & 1) * kWpRowBytes; wprev = wrows + ((y & 1) ^ 1) * kWpRowBytes;
        wp.teN = LdG(reinterpret_cast<const int32_t*>(wprev + 16));
        wp.eA = LdG(reinterpret_cast<const int32_t*>(wprev + (w > 1 ? kWpRecBytes : 0) + 16));
        if constexpr (QUAD) {
          wp.aN[0] = LdG(reinterpret_cast<const uint32_t*>(wprev) + q);
          wp.mA[0] = LdG(reinterpret_cast<const uint32_t*>(wprev + (w > 1 ? kWpRecBytes : 0)) + q);
          wp.aNW[0] = wp.aN[0];
        } else {
          const uint4 m0 = LdG(reinterpret_cast<const uint4*>(wprev));
          const uint4 m1 = LdG(reinterpret_cast<const uint4*>(wprev + (w > 1 ? kWpRecBytes : 0)));
          wp.aN[0] = m0.x; wp.aN[1] = m0.y; wp.aN[2] = m0.z; wp.aN[3] = m0.w;
          wp.mA[0] = m1.x; wp.mA[1] = m1.y; wp.mA[2] = m1.z; wp.mA[3] = m1.w;
#pragma unroll
          for (int i = 0; i < 4; i++) wp.aNW[i] = wp.aN[i];
        }
        wp.teNW = wp.teN; wp.teW = 0;
      }
      // everything this branch loaded has arrived before the branch ends: the common path below then starts without a wait for it
      asm volatile("" : "+v"(up0), "+v"(up1), "+v"(up2), "+v"(const_cluster));
    }
    // ---- one sample.  All loads that do not depend on this sample's context go out first (bit-stream refill, the sample three
    // positions ahead in the row above, the weighted-predictor record two positions ahead); the two that do (cluster, alias entry)
    // are the iteration's two L2 round trips.
    br.Refill();
    int32_t up3 = 0;
    {
      const uint32_t j = x + 3;
      const bool nxt = j >= w;
      if ((nxt ? roll_next : hp) && !(xp & 4)) up3 = LdG(out + (nxt ? (int32_t)(j - w) : (int32_t)j - stride));   // row above, or the start of this row for the row below
    }
    const int32_t n_raw = up0;
    const int32_t W = x ? left : (hp ? n_raw : 0);
    const int32_t N = hp ? n_raw : W;
    const int32_t NW = (x && hp) ? nw : W;
    nw = n_raw;
    const int32_t grad = (int32_t)((uint32_t)W + (uint32_t)N - (uint32_t)NW);
    const bool wp_live = WP && (cls & kLfSimtWpLive);
    const bool last_col = x + 1 >= w;
    const int32_t NE = GEN ? ((hp && !last_col) ? up1 : N) : 0;
    int32_t wp_err = 0, teNE = 0;
    uint32_t aNE[4] = {0, 0, 0, 0};
    if (wp_live) {
      const uint8_t* r = wprev + min(x + 2, w - 1) * kWpRecBytes;
      wp.eB = LdG(reinterpret_cast<const int32_t*>(r + 16));
      if constexpr (QUAD) {
        wp.mB[0] = LdG(reinterpret_cast<const uint32_t*>(r) + q);
        aNE[0] = last_col ? wp.aN[0] : wp.mA[0];
      } else {
        const uint4 m = LdG(reinterpret_cast<const uint4*>(r));
        wp.mB[0] = m.x; wp.mB[1] = m.y; wp.mB[2] = m.z; wp.mB[3] = m.w;
#pragma unroll
        for (int i = 0; i < 4; i++) aNE[i] = last_col ? wp.aN[i] : wp.mA[i];
      }
      teNE = last_col ? wp.teN : wp.eA;
      int32_t p = wp.teW;
      if (abs(wp.teN) > abs(p)) p = wp.teN;
      if (abs(wp.teNW) > abs(p)) p = wp.teNW;
      if (abs(teNE) > abs(p)) p = teNE;
      wp_err = p;
    }
    const uint32_t kind = cls & 3, pcode = (cls >> 2) & 7;
    uint32_t cluster = const_cluster;
    if (kind) {
      uint32_t idx = (uint32_t)(min(max(grad, -512), 511) + 512);
      if (GEN) {
        const uint32_t sa = (cls >> 5) & 3, sb = (cls >> 7) & 3;
        const int32_t va = sa == 0 ? grad : sa == 1 ? W : sa == 2 ? N : wp_err;
        const int32_t vb = sb == 0 ? grad : sb == 1 ? W : sb == 2 ? N : wp_err;
        const uint32_t i1 = (uint32_t)(min(max(va, -512), 511) + 512);
        const uint32_t i2 = ((uint32_t)(min(max(va, -16), 15) + 16) << 5) | (uint32_t)(min(max(vb, -16), 15) + 16);
        idx = kind == 1 ? i1 : i2;
      }
      cluster = LdG(luts + lut_off + idx);
    }
    const int32_t mn = min(N, W), mx = max(N, W);
    const int32_t g5 = NW < mn ? mx : (NW > mx ? mn : grad);
    int32_t guess = pcode == 0 ? 0 : (pcode == 1 ? W : g5);
    if (GEN) {
      if (pcode == 2) guess = N;
      if (__builtin_expect(pcode >= 5, 0)) {
        if (pcode == 5) guess = (int32_t)(((int64_t)W + N) / 2);
        else if (pcode == 6) { const int64_t pp = (int64_t)W + N - NW; guess = Abs64(pp - W) < Abs64(pp - N) ? W : N; }
        else guess = NE;
      }
    }
    if (QUAD && wp_live) {
      // this lane's sub-predictor: weight from its error magnitudes around the sample, prediction = base - ((error sum x coefficient) >> 5)
      const uint32_t e = wp.aN[0] + aNE[0] + wp.aNW[0];
      const int shift = max(0, 26 - (int)__clz((int)(e + 1)));
      uint32_t wt = 4 + (((q == 0 ? 13u : 12u) * LdG(s_div + (e >> shift))) >> shift);
      const int32_t N8 = N << 3, W8 = W << 3, NE8 = NE << 3;
      const int32_t sumWN = wp.teN + wp.teW;
      const int32_t base = q == 0 ? W8 + NE8 - N8 : (q == 2 ? W8 : N8);
      const int32_t tsum = q == 1 ? sumWN + teNE : (q == 2 ? sumWN + wp.teNW : wp.teNW + wp.teN + teNE);
      const int32_t coef = q == 0 ? 0 : (q == 1 ? 16 : (q == 2 ? 10 : 7));      // ((x * 16) >> 5 == x >> 1: sub-predictor 1)
      wp.pr[0] = base - ((tsum * coef) >> 5);
      uint32_t total = (uint32_t)QuadSum((int32_t)wt);
      const int lg = 31 - (int)__clz((int)total);
      wt >>= (lg - 4);
      total = (uint32_t)QuadSum((int32_t)wt);
      const int32_t acc = QuadSum(wp.pr[0] * (int32_t)wt) + (int32_t)(total >> 1) - 1;
      int32_t avg = (int32_t)(((int64_t)acc * (int64_t)LdG(s_div + (total - 1))) >> 24);
      if (((wp.teN ^ wp.teW) | (wp.teN ^ wp.teNW)) <= 0) avg = max(min(W8, min(NE8, N8)), min(max(W8, max(NE8, N8)), avg));
      wp.avg = avg;
      if (pcode == 4) guess = (avg + 3) >> 3;
    } else if (wp_live) {
      // sub-predictor weights from the error magnitudes around the sample
      uint32_t wt[4];
#pragma unroll
      for (int i = 0; i < 4; i++) {
        const uint32_t e = wp.aN[i] + aNE[i] + wp.aNW[i];
        const int shift = max(0, 26 - (int)__clz((int)(e + 1)));
        wt[i] = 4 + (((i == 0 ? 13u : 12u) * LdG(s_div + (e >> shift))) >> shift);
      }
      const int32_t N8 = N << 3, W8 = W << 3, NE8 = NE << 3;
      const int32_t sumWN = wp.teN + wp.teW;
      wp.pr[0] = W8 + NE8 - N8;
      wp.pr[1] = N8 - ((sumWN + teNE) >> 1);
      wp.pr[2] = W8 - (((sumWN + wp.teNW) * 10) >> 5);
      wp.pr[3] = N8 - (((wp.teNW + wp.teN + teNE) * 7) >> 5);
      uint32_t total = wt[0] + wt[1] + wt[2] + wt[3];
      const int lg = 31 - (int)__clz((int)total);
#pragma unroll
      for (int i = 0; i < 4; i++) wt[i] >>= (lg - 4);
      total = wt[0] + wt[1] + wt[2] + wt[3];
      int32_t acc = (int32_t)(total >> 1) - 1;
#pragma unroll
      for (int i = 0; i < 4; i++) acc += wp.pr[i] * (int32_t)wt[i];
      int32_t avg = (int32_t)(((int64_t)acc * (int64_t)LdG(s_div + (total - 1))) >> 24);
      if (((wp.teN ^ wp.teW) | (wp.teN ^ wp.teNW)) <= 0) avg = max(min(W8, min(NE8, N8)), min(max(W8, max(NE8, N8)), avg));
      wp.avg = avg;
      if (pcode == 4) guess = (avg + 3) >> 3;
    }
    // rANS symbol + hybrid integer (dec_ans.h), tables through the L2
    const uint32_t res = state & 0xFFF;
    const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
    const uint64_t e = LdG(alias + ((cluster << la) + i));
    const uint32_t cfg = cfgu != 0xFFFFFFFFu ? cfgu : LdG(cfgp + cluster);
    const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
    const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
    const bool hit = pos >= cutoff;
    uint32_t tok = hit ? right : i;
    state = (hit ? freq1 : freq0) * (state >> 12) + (hit ? offs1 + pos : pos);
    // (the alias entry has arrived, hence every load issued before it: rotate the window over the row above now, not at the top of the
    // next iteration, where the move would wait for whatever the end of this iteration has in flight)
    up0 = up1; up1 = up2; up2 = up3;
    if (state < (1u << 16)) { state = (state << 16) | (uint32_t)(br.buf & 0xFFFFu); br.buf >>= 16; br.avail -= 16; }   // (>= 33 bits were buffered)
    const uint32_t split_exp = cfg & 0xFF;
    if (tok >= (1u << split_exp)) {
      const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
      const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - (1u << split_exp)) >> (msb + lsb))) & 31;
      const uint32_t low = tok & ((1u << lsb) - 1);
      tok >>= lsb;
      if ((int)nbits > br.avail) br.Refill();
      const uint32_t bits = (uint32_t)(br.buf & ((1ull << nbits) - 1));
      br.buf >>= nbits; br.avail -= (int)nbits;
      const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
      tok = (((hi << nbits) | bits) << lsb) | low;
    }
    const int32_t val = (int32_t)((uint32_t)UnpackSigned(tok) + (uint32_t)guess);
    if (!(xp & 2)) StG(out + x, val);
    left = val;
    if (wp_live) {
      // what the sample turned out to be: true error, error magnitude of every sub-predictor -> this row's record; the chain for x + 1
      const int32_t v8 = (int32_t)((uint32_t)val << 3);
      const int32_t te = wp.avg - v8;
      uint8_t* rec = wcur + x * kWpRecBytes;
      if constexpr (QUAD) {
        const uint32_t em = (uint32_t)(abs(wp.pr[0] - v8) + 3) >> 3;
        StG(reinterpret_cast<uint32_t*>(rec) + q, em);
        StG(reinterpret_cast<int32_t*>(rec + 16), te);
        wp.aNW[0] = wp.aN[0]; wp.aN[0] = wp.mA[0] + em; wp.mA[0] = wp.mB[0];
      } else {
        uint32_t em[4];
#pragma unroll
        for (int i = 0; i < 4; i++) em[i] = (uint32_t)(abs(wp.pr[i] - v8) + 3) >> 3;
        StG(reinterpret_cast<uint4*>(rec), make_uint4(em[0], em[1], em[2], em[3]));
        StG(reinterpret_cast<int32_t*>(rec + 16), te);
#pragma unroll
        for (int i = 0; i < 4; i++) { wp.aNW[i] = wp.aN[i]; wp.aN[i] = wp.mA[i] + em[i]; wp.mA[i] = wp.mB[i]; }
      }
      wp.teNW = wp.teN; wp.teN = wp.eA; wp.eA = wp.eB; wp.teW = te;
      const uint32_t vmax = (xp & 8) ? 16u : (1u << 20);      // (bit 8 of the flags: testing, so that ordinary streams take the hand-back path)
      if (__builtin_expect((uint32_t)val + vmax > 2 * vmax || (uint32_t)te + (1u << 24) > (2u << 24), 0)) {
        // outside the range the 32-bit arithmetic is exact for: the one-wavefront-per-stream kernel decodes this stream again (64-bit)
        const LfSimtStream* st = streams + cur;
        const FrameDev& f = frames[LdG(&st->frame)];
        (f.lf_scratch + (uint64_t)LdG(&st->group) * f.lf_scratch_stride)[2] = kLfRedoMark;
        cur++; c = 7; w = 0; h = 0; x = 0;
        continue;
      }
    }
    x++;
  }
}

// =====================================================================================================================
// K_lfpost: LF dequant (+CfL), adaptive smoothing, LLF (lowest frequencies from LF), per-block EPF sigma
// =====================================================================================================================
__global__ void LfDequantKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || f.use_lf_frame) return;      // (use_lf_frame: f.lf holds the samples of the LF frame, copied there before this stage)
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= f.bw || y >= f.bh) return;
  if (f.subsampled) {
    // compressed_dc.cc DequantDC without chroma-from-luma: every channel on its own grid (packed top-left in the planes)
    for (int c = 0; c < 3; c++) {
      if (x >= (f.bw >> f.hs[c]) || y >= (f.bh >> f.vs[c])) continue;
      const uint32_t gc = ((y << f.vs[c]) / 256) * f.xlfgroups + (x << f.hs[c]) / 256;
      const float mulc = 1.0f / (float)(1 << f.lf_scratch[(uint64_t)gc * f.lf_scratch_stride]);
      const size_t oc = (size_t)y * f.bw + x;
      f.lf[c][oc] = (float)f.lfq[c][oc] * (f.lf_fac[c] * mulc);
    }
    return;
  }
  const uint32_t g = (y / 256) * f.xlfgroups + x / 256;
  const int32_t extra_precision = f.lf_scratch[(uint64_t)g * f.lf_scratch_stride];
  const float mul = 1.0f / (float)(1 << extra_precision);
  const size_t o = (size_t)y * f.bw + x;
  const float vy = (float)f.lfq[1][o] * (f.lf_fac[1] * mul);
  const float vx = (float)f.lfq[0][o] * (f.lf_fac[0] * mul);
  const float vb = (float)f.lfq[2][o] * (f.lf_fac[2] * mul);
  f.lf[1][o] = vy;
  f.lf[0][o] = fmaf(vy, f.cfl_lf_x, vx);
  f.lf[2][o] = fmaf(vy, f.cfl_lf_b, vb);
}

__global__ void LfSmoothKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular) return;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  const int w = (int)f.bw, h = (int)f.bh;
  if (x >= w || y >= h) return;
  const size_t o = (size_t)y * w + x;
  const bool interior = !f.skip_lf_smoothing && w > 2 && h > 2 && x >= 1 && y >= 1 && x + 1 < w && y + 1 < h;
  if (!interior) { for (int c = 0; c < 3; c++) f.lf_tmp[c][o] = f.lf[c][o]; return; }
  const float kW0 = 0.05226273532324128f, kW1 = 0.20345139757231578f, kW2 = 0.0334829185968739f;
  float gap = 0.5f, mc[3], sm[3];
  for (int c = 0; c < 3; c++) {
    const float* t = f.lf[c] + o - w; const float* m = f.lf[c] + o; const float* b = f.lf[c] + o + w;
    const float corner = (t[-1] + t[1]) + (b[-1] + b[1]);
    const float edge = (t[0] + m[-1]) + (m[1] + b[0]);
    mc[c] = m[0];
    sm[c] = fmaf(corner, kW2, fmaf(edge, kW1, mc[c] * kW0));
    gap = fmaxf(gap, fabsf((mc[c] - sm[c]) / f.lf_fac[c]));
  }
  const float factor = fmaxf(0.0f, fmaf(-4.0f, gap, 3.0f));
  for (int c = 0; c < 3; c++) f.lf_tmp[c][o] = fmaf(sm[c] - mc[c], factor, mc[c]);
}

template <int N> __device__ __forceinline__ void FDct1D(float (&v)[N]) {  // unscaled forward DCT (dct-inl.h DCT1DImpl)
  if constexpr (N == 2) { const float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; }
  else if constexpr (N > 2) {
    constexpr int H = N / 2;
    float e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; i++) e[i] = v[i] + v[N - 1 - i];
    FDct1D<H>(e);
    constexpr int L = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : 5;
    static_assert(N <= 32, "FDct1D: LLF blocks are at most 32 wide");
#pragma unroll
    for (int i = 0; i < H; i++) o[i] = (v[i] - v[N - 1 - i]) * d_wc[L][i];
    FDct1D<H>(o);
    o[0] = fmaf(o[0], 1.41421356237309504880f, o[1]);
#pragma unroll
    for (int i = 1; i + 1 < H; i++) o[i] = o[i] + o[i + 1];
#pragma unroll
    for (int i = 0; i < H; i++) { v[2 * i] = e[i]; v[2 * i + 1] = o[i]; }
  }
}
__device__ void FDctDyn(float* v, int n) {  // n in {1,2,4,8}
  if (n == 2) { float t[2] = {v[0], v[1]}; FDct1D<2>(t); v[0] = t[0]; v[1] = t[1]; }
  else if (n == 4) { float t[4]; for (int i = 0; i < 4; i++) t[i] = v[i]; FDct1D<4>(t); for (int i = 0; i < 4; i++) v[i] = t[i]; }
  else if (n == 8) { float t[8]; for (int i = 0; i < 8; i++) t[i] = v[i]; FDct1D<8>(t); for (int i = 0; i < 8; i++) v[i] = t[i]; }
}
__device__ __forceinline__ int Log2Small(int n) { return n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : 3; }

// LLF coefficients of one varblock of CX x CY blocks: scaled forward DCT of its LF samples, columns first
// (dec_group.cc / dct_scales.h)
template <int CX, int CY> __device__ __forceinline__ void LlfBlock(const FrameDev& f, size_t o) {
  constexpr float sr = 1.0f / (float)CY, sc = 1.0f / (float)CX;
  constexpr int lx = CX == 1 ? 0 : CX == 2 ? 1 : CX == 4 ? 2 : 3, ly = CY == 1 ? 0 : CY == 2 ? 1 : CY == 4 ? 2 : 3;
#pragma unroll 1
  for (int c = 0; c < 3; c++) {
    float buf[CY][CX];
    const float* src = f.lf_tmp[c] + o;
#pragma unroll
    for (int xx = 0; xx < CX; xx++) {
      float col[CY];
#pragma unroll
      for (int yy = 0; yy < CY; yy++) col[yy] = src[(size_t)yy * f.bw + xx];
      FDct1D<CY>(col);
#pragma unroll
      for (int v = 0; v < CY; v++) buf[v][xx] = col[v] * sr;
    }
#pragma unroll
    for (int v = 0; v < CY; v++) {
      float row[CX];
#pragma unroll
      for (int u = 0; u < CX; u++) row[u] = buf[v][u];
      FDct1D<CX>(row);
#pragma unroll
      for (int u = 0; u < CX; u++) f.llf[c][o + (size_t)v * f.bw + u] = ((row[u] * sc) * d_resample[ly][v]) * d_resample[lx][u];
    }
  }
}
// one thread per 8x8 block: EPF inverse sigma; blocks that are a whole varblock copy their LF sample as the LLF coefficient
__global__ void LlfSigmaKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || FrameFailed(f)) return;
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= f.bw || y >= f.bh) return;
  const size_t o = (size_t)y * f.bw + x;
  const uint32_t info = f.blk_info[o];
  {  // epf.cc ComputeSigma
    const float sigma_quant = f.epf_quant_mul / ((f.epf_quant_scale * (float)BI_HfMul(info)) * -1.1715728752538099024f);
    float sigma = sigma_quant * f.epf_sharp_lut[BI_Sharp(info)];
    sigma = fminf(-1e-4f, sigma);
    f.inv_sigma[o] = 1.0f / sigma;
  }
  const uint32_t s = BI_Strategy(info);
  if (CoveredX(s) == 1 && CoveredY(s) == 1) { for (int c = 0; c < 3; c++) f.llf[c][o] = f.lf_tmp[c][o]; }
}
// LLF coefficients of the larger varblocks, one thread per entry of the per-group varblock lists (every lane has a varblock;
// one thread per 8x8 block left 90 % of the lanes idle while the others ran the transforms of half a dozen shapes one after
// the other).  One instantiation per shape: static loops, everything in registers.
__global__ __launch_bounds__(256) void LlfKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || FrameFailed(f)) return;
  const uint32_t g = blockIdx.x;
  if (g >= f.num_groups) return;
  const uint32_t bx0 = (g % f.xgroups) * 32, by0 = (g / f.xgroups) * 32;
  const uint32_t count = f.vb_count[g];
  for (uint32_t e = threadIdx.x; e < count; e += blockDim.x) {
    const uint32_t ex = f.vb_list[(size_t)g * 1024 + e].x;
    const uint32_t s = ex & 31, cx = CoveredX(s), cy = CoveredY(s);
    if (cx > 8 || cy > 8 || (cx == 1 && cy == 1)) continue;   // BigIdctKernel derives the LLF of DCT128/256 varblocks itself
    const size_t o = (size_t)(by0 + ((ex >> 21) & 31)) * f.bw + bx0 + ((ex >> 16) & 31);
    switch (cy * 16 + cx) {
      case 0x12: LlfBlock<2, 1>(f, o); break;
      case 0x21: LlfBlock<1, 2>(f, o); break;
      case 0x22: LlfBlock<2, 2>(f, o); break;
      case 0x14: LlfBlock<4, 1>(f, o); break;
      case 0x41: LlfBlock<1, 4>(f, o); break;
      case 0x24: LlfBlock<4, 2>(f, o); break;
      case 0x42: LlfBlock<2, 4>(f, o); break;
      case 0x44: LlfBlock<4, 4>(f, o); break;
      case 0x48: LlfBlock<8, 4>(f, o); break;
      case 0x84: LlfBlock<4, 8>(f, o); break;
      case 0x88: LlfBlock<8, 8>(f, o); break;
      default: break;   // (no other shape among the 27 strategies)
    }
  }
}

// =====================================================================================================================
// K_hf: one decode thread per 256x256 group — ANS coefficient decode (dec_group.cc DecodeACVarBlock)
// =====================================================================================================================
__device__ static const uint8_t kFreqCtx[64] = {0,  0,  1,  2,  3,  4,  5,  6,  7,  8,  9,  10, 11, 12, 13, 14, 15, 15, 16, 16, 17, 17,
                                                18, 18, 19, 19, 20, 20, 21, 21, 22, 22, 23, 23, 23, 23, 24, 24, 24, 24, 25, 25, 25, 25,
                                                26, 26, 26, 26, 27, 27, 27, 27, 28, 28, 28, 28, 29, 29, 29, 29, 30, 30, 30, 30};
__device__ static const uint8_t kNzCtx[64] = {0,   0,   31,  62,  62,  93,  93,  93,  93,  123, 123, 123, 123, 152, 152, 152,
                                              152, 152, 152, 152, 152, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180, 180,
                                              180, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206,
                                              206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206, 206};

// only_prefix: the launch beside the SIMT kernel that takes the frames whose AC code is a prefix (Huffman) code — what cjxl's fast efforts
// write; symbols are then read bit by bit through the canonical-code tables in global memory (jxl_dev.h ReadSymbol)
__global__ __launch_bounds__(512) void HfDecodeKernel(const FrameDev* __restrict__ frames, int lane_stride, uint32_t lds_bytes, int only_prefix) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || FrameFailed(f)) return;
  const bool pfx = f.ac_code.use_prefix != 0, lz77 = f.ac_code.lz77 != 0;
  const bool slow = pfx || lz77;                                       // symbols through the general reader (tables in global memory)
  if (only_prefix && !slow) return;
  if (only_prefix && (f.num_passes != 1 || f.subsampled)) return;      // (progressive / chroma-subsampled prefix- or LZ77-coded frames: HfDecodeSimtKernel walks them)
  // ---- stage the AC entropy code (cfg, context map, alias tables if they fit) and the two context LUTs into LDS
  FastCode code;
  if (threadIdx.x < 64) { StS<uint8_t>(threadIdx.x, kNzCtx[threadIdx.x]); StS<uint8_t>(64 + threadIdx.x, kFreqCtx[threadIdx.x]); }
  StageCode(f.ac_code, code, 128, lds_bytes > 128 ? lds_bytes - 128 : 0, /*with_ctx_map=*/true);
  __syncthreads();
  const int tid = blockIdx.x * blockDim.x + threadIdx.x;
  if (tid % lane_stride) return;
  const uint32_t g = tid / lane_stride;
  if (g >= f.num_groups) return;
  const uint32_t gx = g % f.xgroups, gy = g / f.xgroups;
  const uint32_t bx0 = gx * 32, by0 = gy * 32;
  const uint32_t gbw = min(32u, f.bw - bx0), gbh = min(32u, f.bh - by0);
  BitReaderP br;
  uint64_t limit;
  if (f.single_section) { br.Init(f.cs, f.hf_start_bitpos, f.cs_size); limit = f.cs_size * 8; }
  else {
    const uint32_t si = 2 + f.num_lf_groups + g; const uint64_t off = f.sec_off[si];
    if (off + f.sec_size[si] > f.cs_size) return;      // input that ends inside the frame (progressive flush): this group's stream is not there yet, the group is drawn from its LF part
    br.Init(f.cs, off * 8, off + f.sec_size[si]); limit = (off + f.sec_size[si]) * 8;
  }
  if (f.num_passes != 1) { SetError(f, kErrUnsupported); return; }   // progressive frames go through the SIMT kernel (the host forces it)
  const BlockCtxDev& bcm = *f.bcm;
  const uint32_t nctx = bcm.num_ctxs;
  const uint32_t preset = f.preset_bits ? br.Read((int)f.preset_bits) : 0;
  if (preset >= f.num_hf_presets) { SetError(f, kErrBadValue); return; }
  const uint32_t ctx_offset = 495u * nctx * preset;
  uint32_t state = pfx ? 0x130000u : br.Read(32);
  Lz77State lz;                                      // LZ77 over the values of this group stream (dec_ans.h; distance multiplier 0: no special distances)
  lz.Init(lz77 ? f.lz_ac_window + (size_t)g * kAcLzWindow : nullptr, 0, kAcLzWindow);
  if (lz77 && !f.lz_ac_window) { SetError(f, kErrUnsupported); return; }
  auto read_hybrid = [&](uint32_t ctx) -> uint32_t {
    if (!slow) return FastHybrid(br, state, code, code.Cluster(ctx));
    AnsReader ans; ans.state = state;
    uint32_t v;
    if (!lz77) { const uint32_t cl = LdG(f.ac_code.ctx_map + ctx); v = HybridFromToken(br, LdG(f.ac_code.cfg + cl), ReadSymbol(br, ans, f.ac_code, cl)); }
    else v = Lz77Read(br, lz, ctx, f.ac_code.num_ctx, f.ac_code.lz_min_symbol, f.ac_code.lz_min_length, f.ac_code.lz_len_cfg,
                      [&](uint32_t c) { return (uint32_t)LdG(f.ac_code.ctx_map + c); }, [&](uint32_t cl) { return ReadSymbol(br, ans, f.ac_code, cl); },
                      [&](uint32_t cl) { return LdG(f.ac_code.cfg + cl); });
    state = ans.state;
    return v;
  };
  uint32_t nzrow[3][8];   // 32 bytes per channel packed in 8 words (kept in registers)
#pragma unroll
  for (int c = 0; c < 3; c++)
#pragma unroll
    for (int i = 0; i < 8; i++) nzrow[c][i] = 0;
  auto nz_get = [&](int c, uint32_t x) -> uint32_t {
    uint32_t w = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) if ((x >> 2) == (uint32_t)i) w = nzrow[c][i];
    return (w >> ((x & 3) * 8)) & 0xFF;
  };
  auto nz_set = [&](int c, uint32_t x, uint32_t v) {
#pragma unroll
    for (int i = 0; i < 8; i++) if ((x >> 2) == (uint32_t)i) nzrow[c][i] = (nzrow[c][i] & ~(0xFFu << ((x & 3) * 8))) | (v << ((x & 3) * 8));
  };
  int32_t* cbase[3] = {f.coeff[0] + (size_t)g * 65536, f.coeff[1] + (size_t)g * 65536, f.coeff[2] + (size_t)g * 65536};
  uint32_t nz_written = 0;                           // (bench accounting: coefficients this stream writes)
  for (uint32_t by = 0; by < gbh; by++) {
    for (uint32_t bx = 0; bx < gbw; bx++) {
      const size_t o = (size_t)(by0 + by) * f.bw + bx0 + bx;
      const uint32_t info = LdG(f.blk_info + o);
      if (!BI_First(info)) continue;
      const uint32_t s = BI_Strategy(info);
      const uint32_t cx = CoveredX(s), cy = CoveredY(s), covered = cx * cy;
      const uint32_t l2 = 31 - __clz((int)covered), size = covered * 64, ord = OrderBucket(s);
      const uint32_t qf = BI_HfMul(info);
      uint32_t qf_idx = 0;
      for (uint32_t i = 0; i < bcm.n_qf_thr; i++) qf_idx += qf > bcm.qf_thr[i];
      uint32_t lf_idx = 0;
      if (bcm.num_lf_ctxs > 1 && !f.use_lf_frame) {
        uint32_t b0 = 0, b1 = 0, b2 = 0;
        for (uint32_t i = 0; i < bcm.n_lf_thr[0]; i++) b0 += f.lfq[0][o] > bcm.lf_thr[0][i];
        for (uint32_t i = 0; i < bcm.n_lf_thr[1]; i++) b1 += f.lfq[1][o] > bcm.lf_thr[1][i];
        for (uint32_t i = 0; i < bcm.n_lf_thr[2]; i++) b2 += f.lfq[2][o] > bcm.lf_thr[2][i];
        lf_idx = (b0 * (bcm.n_lf_thr[2] + 1) + b2) * (bcm.n_lf_thr[1] + 1) + b1;
      }
      const uint32_t coff = f.coef_off[o];
#pragma unroll 1
      for (int ci = 0; ci < 3; ci++) {
        const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;  // Y, X, B
        uint32_t idx = (uint32_t)(c < 2 ? (c ^ 1) : 2) * 13 + ord;
        idx = idx * (bcm.n_qf_thr + 1) + qf_idx;
        idx = idx * bcm.num_lf_ctxs + lf_idx;
        const uint32_t block_ctx = bcm.ctx_map[idx];
        uint32_t pred;
        if (bx == 0) pred = by == 0 ? 32 : nz_get(c, bx);
        else if (by == 0) pred = nz_get(c, bx - 1);
        else pred = (nz_get(c, bx) + nz_get(c, bx - 1) + 1) / 2;
        const uint32_t pc = pred > 64 ? 64 : pred;
        const uint32_t nz_ctx = ctx_offset + (pc < 8 ? block_ctx + nctx * pc : block_ctx + nctx * (4 + pc / 2));
        uint32_t nzeros = read_hybrid(nz_ctx);
        if (nzeros + covered > size) { SetError(f, kErrNzeros); return; }
        nz_written += nzeros;
        const uint32_t nzm = (nzeros + covered - 1) >> l2;
        for (uint32_t ix = 0; ix < cx; ix++) nz_set(c, bx + ix, nzm);
        const uint32_t histo = ctx_offset + 37 * nctx + 458 * block_ctx;
        const uint16_t* order = f.orders[ord * 3 + c];
        int32_t* blk = cbase[c] + coff;
        uint32_t prev = nzeros > size / 16 ? 0 : 1;
        uint32_t next_pos = LdG(order + covered);         // coefficient position for k, loaded one step ahead
        for (uint32_t k = covered; k < size && nzeros != 0; k++) {
          const uint32_t pos = next_pos;
          next_pos = k + 1 < size ? LdG(order + k + 1) : 0;
          const uint32_t nzl = (nzeros + covered - 1) >> l2, kk = k >> l2;
          const uint32_t ctx = histo + ((uint32_t)LdS<uint8_t>(nzl) + LdS<uint8_t>(64 + kk)) * 2 + prev;
          const uint32_t u = read_hybrid(ctx);
          prev = u != 0;
          nzeros -= prev;
          if (u) StG(blk + pos, UnpackSigned(u));
        }
        if (nzeros != 0) { SetError(f, kErrNzeros); return; }
      }
    }
  }
  if (state != 0x130000u) { SetError(f, kErrAnsFinalState); return; }
  if (br.BitPos() > limit) { SetError(f, kErrOverrun); return; }
  if (f.hf_end_bitpos) f.hf_end_bitpos[g] = br.BitPos();
  if (nz_written) atomicAdd(f.hf_written, nz_written);
}

// ---- SIMT variant: every lane decodes the stream of its own group; one token per lane per loop iteration -------------
// The per-token work (context -> cluster -> alias lookup -> state update -> refill -> hybrid bits -> store) is the same
// straight-line code for the "number of non-zeros" token and the coefficient tokens, so lanes stay converged.  Nothing in
// the loop waits on global memory: tables, the per-lane bit-stream ring and the non-zero row buffers live in LDS, the
// varblock list entry of the next block and the next coefficient position are loaded one step ahead, and the bit-stream
// rings are topped up every fourth iteration with loads issued four iterations earlier.
constexpr uint32_t kSimtMaxThreads = 256;                              // lanes (= group streams) per workgroup: 64 .. 256, right-sized per batch
constexpr uint32_t kSimtBcmOff = 128;                                  // [0, 128): the two 64-entry context tables; then a copy of the BlockCtxDev
constexpr uint32_t kSimtOrdOff = (kSimtBcmOff + (uint32_t)sizeof(BlockCtxDev) + 15) & ~15u;   // 39 order-table pointers of the pass
constexpr uint32_t kSimtCodeOff = kSimtOrdOff + 320;                   // the AC code of the pass (cfg, context map, alias tables)
constexpr uint32_t kSimtLaneBytes = 96 + 64;                           // per lane, after the code: nzeros row buffers (96 B) + 16-word bit-stream ring
struct BitReaderRing {   // per-lane ring of 16 words in LDS; absolute word index w lives at slot w & 15
  uint32_t wpos, ring_off;
  uint64_t buf;
  int avail;
  __device__ __forceinline__ void Refill() {
    if (avail <= 32) {
      buf |= (uint64_t)LdS<uint32_t>(ring_off + ((wpos & 15) << 2)) << avail;
      avail += 32;
      wpos++;
    }
  }
  __device__ __forceinline__ uint32_t Read(int n) {
    Refill();
    const uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; avail -= n;
    return v;
  }
  __device__ __forceinline__ uint64_t BitPos() const { return (uint64_t)wpos * 32 - (uint64_t)avail; }
};

template <bool ALL_LDS, typename BR> __device__ __forceinline__ uint32_t HybridSimt(BR& br, uint32_t& state, const FastCode& c, uint32_t ctx) {
  br.Refill();   // one refill point per token instead of one per field
  const uint32_t cluster = ALL_LDS ? LdS<uint8_t>(c.ctx_map_off + ctx) : c.Cluster(ctx);
  const uint32_t la = c.log_alpha;
  const uint32_t cfg = ALL_LDS ? LdS<uint32_t>(c.cfg_off + cluster * 4) : c.Cfg(cluster);
  const uint32_t res = state & 0xFFF;
  const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
  uint32_t tok, off, freq;
#ifndef JXL_HF_WIDE_ALIAS
  if (ALL_LDS) {      // StageCodeCompact: 4-byte slot, frequency by decoded symbol
    const uint32_t cb = cluster << la;
    const uint32_t e = LdS<uint32_t>(c.alias_off + ((cb + i) << 2));
    const uint32_t cutoff = e & 0xFF, right = (e >> 8) & 0xFF, offs1 = e >> 16;
    const bool hit = pos >= cutoff;
    tok = hit ? right : i;
    off = hit ? offs1 + pos : pos;
    freq = LdS<uint16_t>(c.freq_off + ((cb + tok) << 1));
  } else
#endif
  {
    const uint64_t e = ALL_LDS ? LdS<uint64_t>(c.alias_off + (((cluster << la) + i) << 3)) : c.Alias(cluster, i);
    const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
    const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
    const bool hit = pos >= cutoff;
    tok = hit ? right : i;
    off = hit ? offs1 + pos : pos;
    freq = hit ? freq1 : freq0;
  }
  state = freq * (state >> 12) + off;
  // (the caller refilled: at least 33 bits are buffered, 17 after the renormalisation — enough for most extra-bit fields)
  if (state < (1u << 16)) { state = (state << 16) | (uint32_t)(br.buf & 0xFFFFu); br.buf >>= 16; br.avail -= 16; }
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  const uint32_t split = 1u << split_exp;
  if (tok < split) return tok;
  uint32_t nbits = split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb));
  nbits &= 31;
  const uint32_t low = tok & ((1u << lsb) - 1);
  tok >>= lsb;
  if ((int)nbits > br.avail) br.Refill();
  const uint32_t bits = (uint32_t)(br.buf & ((1ull << nbits) - 1));
  br.buf >>= nbits; br.avail -= (int)nbits;
  const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
  return (((hi << nbits) | bits) << lsb) | low;
}

template <bool ALL_LDS, bool SUB, bool MULTI> __global__ __launch_bounds__(1024) void HfDecodeSimtKernel(const FrameDev* __restrict__ frames, uint32_t lds_bytes, uint32_t lanes, uint32_t lanes_per_wave,
                                                                                 uint32_t* __restrict__ sync, uint32_t epoch, int prio) {
  // sync[0]: workgroups of this launch that have started, sync[1]: number of the last HF launch whose workgroups all have
  // (the LF stage of a later batch waits for that before its own workgroups are dispatched, see HeadStartKernel)
  // (a counter per launch — slot 2 + epoch % 16 —: up to three HF launches of different sizes run at the same time, and one shared counter could pass both targets
  // without matching either, never reset, and every later LF head start would run into its 2 ms timeout: ADVICE r5)
  if (sync && threadIdx.x == 0) { uint32_t* ctr = sync + 2 + (epoch & 15u); if (atomicAdd(ctr, 1u) + 1 == gridDim.x * gridDim.y) { atomicExch(ctr, 0u); __threadfence(); atomicMax(sync + 1, epoch); } }
  const FrameDev& f = frames[blockIdx.y];
  // (the status word is read once per workgroup — another workgroup of the launch may set it at any time, and wavefronts that disagreed about it
  // would part ways in front of the barriers below; the 8 spare bytes behind the 39 order pointers carry it)
  if (threadIdx.x == 0) StS<uint32_t>(kSimtOrdOff + 312, __hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
  __syncthreads();
  if (f.is_modular || LdS<uint32_t>(kSimtOrdOff + 312) != 0) return;
  // prefix-coded frames: HfDecodeKernel, launched beside this one — unless they are progressive or chroma-subsampled: those are walked here (the
  // general instantiation), their symbols read bit by bit through the canonical-code tables in global memory (jxl_dev.h ReadSymbol)
  bool any_pfx = f.ac_code.use_prefix != 0 || f.ac_code.lz77 != 0;     // (prefix codes or LZ77: the general symbol reader)
  for (uint32_t ps = 1; MULTI && ps < f.num_passes; ps++) any_pfx |= f.passes[ps].code.use_prefix != 0 || f.passes[ps].code.lz77 != 0;
  if (any_pfx && !((SUB && f.subsampled) || (MULTI && f.num_passes > 1))) return;
  if (blockIdx.x * lanes >= f.num_groups) return;
  if (prio & 1) __builtin_amdgcn_s_setprio(3);   // few, long, latency-critical waves on the critical path of a batch: issue ahead of co-resident waves
  // per-lane regions sit at the end of the dynamic LDS: `lanes` real ones + one scratch region that all stream-less
  // lanes of the last wavefront share (they only ever write zeros / prefetched words there and read nothing back)
  const uint32_t lane_off = lds_bytes - (lanes + 1) * kSimtLaneBytes;
  FastCode code;
  if (threadIdx.x < 64) { StS<uint8_t>(threadIdx.x, kNzCtx[threadIdx.x]); StS<uint8_t>(64 + threadIdx.x, kFreqCtx[threadIdx.x]); }
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(f.bcm);
    for (uint32_t i = threadIdx.x; i < sizeof(BlockCtxDev) / 4; i += blockDim.x) StS<uint32_t>(kSimtBcmOff + i * 4, LdG(src + i));
  }
  int32_t* const cbase0 = f.coeff[0]; int32_t* const cbase1 = f.coeff[1]; int32_t* const cbase2 = f.coeff[2];
  // `lanes` group streams per workgroup, `lanes_per_wave` of them on the first lanes of each wavefront.  Sparse wavefronts
  // on purpose: every lane state that any lane of a wavefront is in costs the whole wavefront its instructions (block
  // start, nzeros token, coefficient token, bit-stream refills ...), and with 64 streams per wavefront every iteration
  // runs every path; 16 streams per wavefront skip most of them, and the 2-3 wavefronts a SIMD then holds fill each
  // other's issue gaps (the per-lane LDS regions — the binding resource — do not change).
  const uint32_t slot = (threadIdx.x >> 6) * lanes_per_wave + (threadIdx.x & 63);
  const uint32_t g = blockIdx.x * lanes + slot;
  bool dead = (threadIdx.x & 63) >= lanes_per_wave || slot >= lanes || g >= f.num_groups;   // no stream, or a stream that failed in an earlier pass
  const uint32_t lane_slot = dead ? lanes : slot;
  const uint32_t nz_base = lane_off + lane_slot * 96;
  const uint32_t wend_all = (uint32_t)((f.cs_size + 3) >> 2);   // 16-byte loads stay inside the codestream buffer
  // Progressive frames: PassGroup section (pass, g) carries value >> shift of every coefficient under the pass's own code
  // and orders; the values accumulate.  The tables are re-staged per pass (block-wide), the lanes restart their streams.
  // MULTI = false: a launch without progressive frames — one pass, no shift, no accumulation in the loop
  for (uint32_t pass = 0; pass < (MULTI ? f.num_passes : 1u); pass++) {
  const PassDev& pd = f.passes[pass];
  const uint32_t shift = MULTI ? pd.shift : 0u;
  if (pass) __syncthreads();                           // every lane is done with the previous pass's tables
#ifndef JXL_HF_WIDE_ALIAS
  if (ALL_LDS && !(pd.code.use_prefix != 0 || pd.code.lz77 != 0)) StageCodeCompact(pd.code, code, kSimtCodeOff, lane_off > kSimtCodeOff ? lane_off - kSimtCodeOff : 0);
  else
#endif
  StageCode(pd.code, code, kSimtCodeOff, lane_off > kSimtCodeOff ? lane_off - kSimtCodeOff : 0, /*with_ctx_map=*/true);
  if (threadIdx.x < 39) StS<uint64_t>(kSimtOrdOff + threadIdx.x * 8, (uint64_t)(uintptr_t)pd.orders[threadIdx.x]);
  __syncthreads();
  const bool lz77 = (SUB || MULTI) && pd.code.lz77 != 0;
  const bool pfx = (SUB || MULTI) && (pd.code.use_prefix != 0 || lz77);      // "slow" pass: symbols through the general reader
  Lz77State lz;                                        // (one window per group stream, restarted pass after pass)
  if (SUB || MULTI) lz.Init(lz77 && !dead ? f.lz_ac_window + (size_t)g * kAcLzWindow : nullptr, 0, kAcLzWindow);
  if (lz77 && !f.lz_ac_window) { if (threadIdx.x == 0) SetError(f, kErrUnsupported); return; }
  if (ALL_LDS && !pfx && (code.cfg_off == kNotInLds || code.ctx_map_off == kNotInLds || code.alias_off == kNotInLds)) { if (threadIdx.x == 0) SetError(f, kErrUnsupported); return; }
  bool done = dead;
  uint32_t err = 0;                                    // first error of this lane's stream (reported after the loop)
  for (uint32_t i = 0; i < 24; i++) StS<uint32_t>(nz_base + i * 4, 0u);
  constexpr uint32_t oMap = kSimtBcmOff + offsetof(BlockCtxDev, ctx_map);
  const uint32_t n_qf = LdS<uint32_t>(kSimtBcmOff + offsetof(BlockCtxDev, n_qf_thr));
  const uint32_t num_lf_ctxs = LdS<uint32_t>(kSimtBcmOff + offsetof(BlockCtxDev, num_lf_ctxs));
  const uint32_t nctx = LdS<uint32_t>(kSimtBcmOff + offsetof(BlockCtxDev, num_ctxs));
  const uint32_t qlf_stride = (n_qf + 1) * num_lf_ctxs;   // the LF stage stored qf_idx * num_lf_ctxs + lf_idx per varblock
  const uint32_t gsafe = done ? 0 : g;
  // ---- per-lane bit-stream ring
  uint64_t bit0, byte_end;
  if (f.single_section) { bit0 = f.hf_start_bitpos; byte_end = f.cs_size; }
  else { const uint32_t si = 2 + f.num_lf_groups + pass * f.num_groups + gsafe; const uint64_t off = LdG(f.sec_off + si), sz = LdG(f.sec_size + si); bit0 = off * 8; byte_end = off + sz; }
  // input that ends inside the frame (progressive flush, host_parse.cc FramePlan::partial): a group stream that is not completely there is left out, and with it the
  // group's later passes — the group is drawn from what has arrived (dec_frame.cc Flush).  Complete frames never get here: their sections lie inside the codestream.
  if (!done && byte_end > f.cs_size) { done = true; dead = true; bit0 = 0; }
  const uint64_t limit = byte_end * 8;
  const uint32_t* words = reinterpret_cast<const uint32_t*>(f.cs);
  BitReaderRing br;
  br.ring_off = lane_off + (lanes + 1) * 96 + lane_slot * 64;
  br.wpos = (uint32_t)(bit0 >> 5);
  br.buf = 0; br.avail = 0;
  uint32_t wload = br.wpos & ~3u;             // next absolute word index to fetch (multiple of 4: 16-byte loads)
  // (words past the end of a section are the next section's, not zeros: a valid stream never consumes them — the ring only
  // prefetches them — and an invalid one fails the final-state / overrun checks either way)
  // (the codestream is followed by 80 zero bytes, host_parse.cc ExtractCodestream: the prefetch of a valid stream ends inside them; a
  // damaged stream that runs past its section keeps re-reading the last words — it fails the final-state / overrun checks either way)
  const uint32_t wmax = (wend_all + 12) & ~3u;
  auto fetch4 = [&](uint32_t w) -> uint4 { return LdG(reinterpret_cast<const uint4*>(words + min(w, wmax))); };
  auto put4 = [&](uint32_t w, const uint4& v) { StS<uint4>(br.ring_off + ((w & 15) << 2), v); };
  for (int i = 0; i < 4; i++) { put4(wload, fetch4(wload)); wload += 4; }   // 16 words ahead
  uint4 pend0 = make_uint4(0, 0, 0, 0), pend1 = pend0;
  bool pending = false;
  {
    br.Refill();
    const int skip = (int)(bit0 & 31);
    br.buf >>= skip; br.avail -= skip;
  }
  uint32_t ctx_offset = 0, state = 0;
  if (!done) {
    const uint32_t preset = f.preset_bits ? br.Read((int)f.preset_bits) : 0;
    if (preset >= f.num_hf_presets) { err = kErrBadValue; done = true; }
    ctx_offset = 495u * nctx * preset;
    state = (pfx && pd.code.use_prefix) ? 0x130000u : br.Read(32);           // (prefix codes carry no ANS state)
  }
  // ---- varblock list of this group, one entry loaded ahead
  const uint2* vbl = f.vb_list + (size_t)gsafe * 1024;
  const uint32_t nvb = done ? 0 : LdG(f.vb_count + gsafe);
  const uint32_t gbase = gsafe * 65536u;
  uint32_t vi = 0;
  uint2 ent_next = nvb ? LdG(vbl) : make_uint2(0, 0);
  // chroma subsampling shifts (0 / 1); SUB = false: a launch without subsampled frames, the lock-step loop does not carry their bookkeeping
  const uint32_t sub_pack = SUB ? (f.hs[0] | (f.vs[0] << 1) | (f.hs[1] << 2) | (f.vs[1] << 3) | (f.hs[2] << 4) | (f.vs[2] << 5)) : 0u;
  uint32_t phase = 1;                     // 1: read nzeros, 2: read a coefficient (varblock starts ride on the iteration that ends the previous one)
  uint32_t bx = 0, by = 0, ci = 0, covered = 1, l2 = 0, size = 64, ord = 0, lcx = 0, coff = 0, qlf = 0;
  uint32_t nzeros = 0, prev = 0, k = 0, histo = 0, next_pos = 0, nz_total = 0, bctx_idx = 0;
  const uint32_t bctx_step = 13u * qlf_stride;
  uint64_t end_bitpos = 0;
  const uint16_t* order = pd.orders[0];
  int32_t* blk = cbase0;
  uint32_t iter = 0;
  // Start of the next varblock, or the end of the stream.  Runs in the iteration that finishes the previous varblock (a lane used to
  // spend an iteration of its own in a "block start" phase without decoding a token: one in about twelve); the entry after the one
  // consumed here is requested right away, so its load has the whole varblock (at least three tokens) to arrive.
  auto block_start = [&]() {
    if (vi >= nvb) {
      if (state != 0x130000u) err = kErrAnsFinalState;
      else if (br.BitPos() > limit) err = kErrOverrun;
      end_bitpos = br.BitPos();
      done = true;
    } else {
      const uint32_t ex = ent_next.x, ey = ent_next.y;
      bx = (ex >> 16) & 31; by = (ex >> 21) & 31; qlf = ex >> 26;
      lcx = (ex >> 5) & 7; l2 = (ex >> 8) & 15; ord = (ex >> 12) & 15;
      covered = 1u << l2; size = covered * 64;
      coff = gbase + ey;
      bctx_idx = ord * qlf_stride + qlf;             // block-context index of channel ci = this + ci * 13 * qlf_stride
      ci = 0; phase = 1;
      vi++;
      if (vi < nvb) ent_next = LdG(vbl + vi);
    }
  };
  if (!done) block_start();
  while (__ballot(!done) != 0ull) {
    // ---- bit-stream top-up, every 2nd iteration (<= 3 words consumed in between): store what was requested 2 iterations
    // ago, request the next 8 words when at most 8 are buffered (ring of 16: never overwritten, never empty)
    if ((iter & 1) == 0) {
      if (pending) { put4(wload - 8, pend0); put4(wload - 4, pend1); pending = false; }
      if (!done && wload - br.wpos <= 8) { pend0 = fetch4(wload); pend1 = fetch4(wload + 4); wload += 8; pending = true; }
    }
    iter++;
    if (!done) {
      const uint32_t c = ci == 0 ? 1 : ci == 1 ? 0 : 2;  // Y, X, B
      // chroma-subsampled frames (dec_group.cc): a channel has a block only where the block starts one of its cells, and its
      // "non-zeros" neighbourhood lives on its own grid (nbx / nby)
      const uint32_t hsc = SUB ? (sub_pack >> (2 * c)) & 1u : 0u, vsc = SUB ? (sub_pack >> (2 * c + 1)) & 1u : 0u;
      const uint32_t nbx = bx >> hsc, nby = by >> vsc;
      // coefficient position after the current one: requested before the token is decoded, as the aligned 32-bit word that
      // holds it (the 16-bit field is only extracted where it is used, so nothing waits for the load up here)
      uint32_t ctx, fetched = 0, fetched_sh = 0;
      auto fetch_pos = [&](uint32_t kk) { fetched = LdG(reinterpret_cast<const uint32_t*>(order + (kk & ~1u))); fetched_sh = (kk & 1u) << 4; };
      if (SUB && phase == 1 && ((bx & hsc) | (by & vsc)) != 0) {
        ci++;                                      // this channel has no block here
        if (ci == 3) block_start(); else phase = 1;
      } else {
      if (phase == 1) {
        order = reinterpret_cast<const uint16_t*>((uintptr_t)LdS<uint64_t>(kSimtOrdOff + (ord * 3 + c) * 8));
        fetch_pos(covered);
        const uint32_t block_ctx = LdS<uint8_t>(oMap + bctx_idx + ci * bctx_step);   // ((c < 2 ? c ^ 1 : 2) == ci) * 13 + ord) * qlf_stride + qlf
        const uint32_t top = LdS<uint8_t>(nz_base + c * 32 + nbx), left = nbx ? LdS<uint8_t>(nz_base + c * 32 + nbx - 1) : 0;
        const uint32_t pred = nbx == 0 ? (nby == 0 ? 32 : top) : nby == 0 ? left : (top + left + 1) / 2;
        const uint32_t pc = pred > 64 ? 64 : pred;
        ctx = ctx_offset + (pc < 8 ? block_ctx + nctx * pc : block_ctx + nctx * (4 + pc / 2));
        histo = ctx_offset + 37 * nctx + 458 * block_ctx;
      } else {
        if (k + 1 < size) fetch_pos(k + 1);
        const uint32_t nzl = (nzeros + covered - 1) >> l2, kk = k >> l2;
        ctx = histo + ((uint32_t)LdS<uint8_t>(nzl) + LdS<uint8_t>(64 + kk)) * 2 + prev;
      }
      uint32_t u;
      if ((SUB || MULTI) && pfx) {
        br.Refill();
        AnsReader ans; ans.state = state;
        if (!lz77) { const uint32_t cl = LdG(pd.code.ctx_map + ctx); u = HybridFromToken(br, LdG(pd.code.cfg + cl), ReadSymbol(br, ans, pd.code, cl)); }
        else u = Lz77Read(br, lz, ctx, pd.code.num_ctx, pd.code.lz_min_symbol, pd.code.lz_min_length, pd.code.lz_len_cfg,
                          [&](uint32_t c2) { return (uint32_t)LdG(pd.code.ctx_map + c2); }, [&](uint32_t cl) { return ReadSymbol(br, ans, pd.code, cl); },
                          [&](uint32_t cl) { return LdG(pd.code.cfg + cl); });
        state = ans.state;
      } else u = HybridSimt<ALL_LDS>(br, state, code, ctx);
      if (phase == 1) {
        nzeros = u;
        nz_total += u;                               // (bench accounting: coefficients this stream writes)
        if (nzeros + covered > size) { err = kErrNzeros; done = true; }
        const uint32_t nzm = (nzeros + covered - 1) >> l2;
        {  // the varblock's columns of the "non-zeros above" row: one masked read-modify-write of the aligned 8 bytes that
           // hold them (varblocks of up to 8 columns sit on multiples of their width); wider or unaligned ones byte by byte
          const uint32_t cxw = 1u << lcx, sh = (nbx & 7) * 8;
          if ((nbx & 7) + cxw <= 8) {
            const uint32_t a8 = nz_base + c * 32 + (nbx & ~7u);
            const uint64_t mask = (cxw == 8 ? ~0ull : ((1ull << (8 * cxw)) - 1ull)) << sh;
            const uint64_t val = (uint64_t)nzm * 0x0101010101010101ull;
            const uint64_t old = LdS<uint64_t>(a8);
            StS<uint64_t>(a8, (old & ~mask) | (val & mask));
          } else {
            for (uint32_t ix = 0; ix < cxw; ix++) StS<uint8_t>(nz_base + c * 32 + nbx + ix, (uint8_t)nzm);
          }
        }
        blk = (c == 0 ? cbase0 : c == 1 ? cbase1 : cbase2) + coff;
        prev = nzeros > size / 16 ? 0 : 1;
        k = covered;
        next_pos = (fetched >> fetched_sh) & 0xFFFFu;
        phase = 2;
      } else {
        const uint32_t pos = next_pos;
        next_pos = (fetched >> fetched_sh) & 0xFFFFu;
        if (u) {
          int32_t val = (int32_t)((uint32_t)UnpackSigned(u) << shift);
          if (MULTI && pass) val = (int32_t)((uint32_t)val + (uint32_t)LdG(blk + pos));
          if (!(prio & 4)) StG(blk + pos, val);       // (JXL_HIP_HF_PRIO bit 2: experiment — what the scattered stores cost the kernels beside this one; wrong pixels)
          else if (prio & 8) StG(blk + (nz_total & 0xFFFFu), val);   // (bit 3: the same number of stores, but consecutive per lane)
        }
        prev = u != 0;
        nzeros -= prev;
        k++;
        if (nzeros != 0 && k >= size) { err = kErrNzeros; done = true; }
      }
      if (phase == 2 && nzeros == 0) { ci++; if (ci == 3) block_start(); else phase = 1; }
      }
    }
  }
  if (err) { SetError(f, err); dead = true; }
  else if (!dead && f.hf_end_bitpos && pass == f.mod_pass) f.hf_end_bitpos[g] = end_bitpos;   // the Modular part of the extra channels' pass follows its coefficients
  if (!dead && nz_total) { atomicAdd(f.hf_written, nz_total); nz_total = 0; }
  }  // passes
}

// ---- wave-wide variant (round 6): ONE group stream per wavefront, every lane running the chain — what a single image (or a handful) needs: 135 streams of a 4K frame
// on 135 wavefronts.  The sparse SIMT form above spends ~220 instructions and several dependent LDS round trips per token on one busy lane (1.4 us per token with the
// block bookkeeping); here the state of the stream (ANS state, bit buffer, position in the block, counts) sits in scalar registers, the bit stream in a VGPR (one word
// per lane, WaveBits), the "non-zeros above" rows in three VGPRs (one block column per lane), the varblock list 64 entries at a time in two, the first 64 entries of
// every coefficient order in LDS (the head of an order table is where most blocks end).  The context of a coefficient token depends on the token before it only through
// "was it zero" and the count of non-zeros left — two outcomes: the even / odd lanes work out the context, its cluster and the cluster's alias-table base for both while
// the token before is still being decoded, and the alias read of the next token goes out with per-lane addresses the moment the ANS state is known; `v_readlane` with the
// outcome picks the winner (StageCode's wide layout, as DecodeChannelWave).  Single-pass ANS-coded frames without chroma subsampling; the host sends everything else
// to the SIMT kernel.
constexpr uint32_t kHwOrdOff = (kSimtBcmOff + (uint32_t)sizeof(BlockCtxDev) + 15) & ~15u;   // 39 x 64 u16: heads of the order tables
constexpr uint32_t kHwCodeOff = kHwOrdOff + 39 * 128;
constexpr uint32_t kHwWaves = 4;
__global__ __launch_bounds__(64 * kHwWaves) void HfDecodeWaveKernel(const FrameDev* __restrict__ frames, uint32_t lds_bytes) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || FrameFailed(f)) return;
  if (blockIdx.x * kHwWaves >= f.num_groups) return;
  const PassDev& pd = f.passes[0];
  FastCode code;
  if (threadIdx.x < 64) { StS<uint8_t>(threadIdx.x, kNzCtx[threadIdx.x]); StS<uint8_t>(64 + threadIdx.x, kFreqCtx[threadIdx.x]); }
  {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(f.bcm);
    for (uint32_t i = threadIdx.x; i < sizeof(BlockCtxDev) / 4; i += blockDim.x) StS<uint32_t>(kSimtBcmOff + i * 4, LdG(src + i));
  }
  for (uint32_t i = threadIdx.x; i < 39 * 64; i += blockDim.x) StS<uint16_t>(kHwOrdOff + i * 2, LdG(pd.orders[i >> 6] + (i & 63)));
  StageCode(pd.code, code, kHwCodeOff, lds_bytes > kHwCodeOff ? lds_bytes - kHwCodeOff : 0, /*with_ctx_map=*/true, /*with_wide=*/true, /*with_plain=*/false);
  __syncthreads();
  if (code.wide_off == kNotInLds || code.ctx_map_off == kNotInLds || code.cfg_off == kNotInLds) { if (threadIdx.x == 0) SetError(f, kErrUnsupported); return; }
  const uint32_t lane = threadIdx.x & 63, g = blockIdx.x * kHwWaves + (threadIdx.x >> 6);
  if (g >= f.num_groups) return;
  __builtin_amdgcn_s_setprio(3);
  constexpr uint32_t oMap = kSimtBcmOff + offsetof(BlockCtxDev, ctx_map);
  const uint32_t n_qf = Uniform(LdS<uint32_t>(kSimtBcmOff + offsetof(BlockCtxDev, n_qf_thr)));
  const uint32_t num_lf_ctxs = Uniform(LdS<uint32_t>(kSimtBcmOff + offsetof(BlockCtxDev, num_lf_ctxs)));
  const uint32_t nctx = Uniform(LdS<uint32_t>(kSimtBcmOff + offsetof(BlockCtxDev, num_ctxs)));
  const uint32_t qlf_stride = (n_qf + 1) * num_lf_ctxs, bctx_step = 13u * qlf_stride;
  const uint32_t la = Uniform(code.log_alpha), cfg_off = Uniform(code.cfg_off), cfg_uniform = Uniform(code.cfg_uniform), map_off = Uniform(code.ctx_map_off);
  const uint32_t wide_off = Uniform(code.wide_off), cut_off = Uniform(code.cut_off);
  uint64_t bit0, byte_end;
  if (f.single_section) { bit0 = f.hf_start_bitpos; byte_end = f.cs_size; }
  else { const uint32_t si = 2 + f.num_lf_groups + g; const uint64_t off = LdG(f.sec_off + si), sz = LdG(f.sec_size + si); bit0 = off * 8; byte_end = off + sz; }
  if (byte_end > f.cs_size) return;      // input that ends inside the frame (progressive flush): the group is drawn from its LF part
  const uint64_t limit = byte_end * 8;
  WaveBits bits;
  bits.Start(reinterpret_cast<const uint32_t*>(f.cs), (uint32_t)((f.cs_size + 3) >> 2) + 16u, ((uint64_t)Uniform((uint32_t)(bit0 >> 32)) << 32) | Uniform((uint32_t)bit0), lane);   // (80 zero bytes follow the codestream)
  const uint32_t preset = f.preset_bits ? bits.Read((int)f.preset_bits) : 0;
  uint32_t err = 0;
  if (preset >= f.num_hf_presets) err = kErrBadValue;
  const uint32_t ctx_offset = 495u * nctx * preset;
  uint32_t state = bits.Read(32);
  const uint2* vbl = f.vb_list + (size_t)g * 1024;
  const uint32_t nvb = err ? 0u : Uniform(LdG(f.vb_count + g));
  const uint32_t gbase = g * 65536u;
  int32_t* const cb0 = f.coeff[0]; int32_t* const cb1 = f.coeff[1]; int32_t* const cb2 = f.coeff[2];
  uint32_t nzr0 = 0, nzr1 = 0, nzr2 = 0;        // "non-zeros" of the blocks above, per channel: lane = block column
  uint32_t entx = 0, enty = 0;
  uint32_t nz_total = 0;
  for (uint32_t vi = 0; vi < nvb && !err; vi++) {
    if ((vi & 63) == 0) {
      const uint2 ent = vi + lane < nvb ? LdG(vbl + vi + lane) : make_uint2(0, 0);
      entx = ent.x; enty = ent.y;
      __builtin_amdgcn_s_waitcnt(0x0F70);
    }
    const uint32_t ex = (uint32_t)__builtin_amdgcn_readlane((int)entx, (int)(vi & 63)), ey = (uint32_t)__builtin_amdgcn_readlane((int)enty, (int)(vi & 63));
    const uint32_t bx = (ex >> 16) & 31, by = (ex >> 21) & 31, qlf = ex >> 26, lcx = (ex >> 5) & 7, l2 = (ex >> 8) & 15, ord = (ex >> 12) & 15;
    const uint32_t covered = 1u << l2, size = covered * 64, coff = gbase + ey;
    const uint32_t bctx_idx = ord * qlf_stride + qlf;
    const bool in_cols = lane >= bx && lane < bx + (1u << lcx);
#pragma unroll
    for (int ci = 0; ci < 3; ci++) {
      if (err) break;
      const int c = ci == 0 ? 1 : ci == 1 ? 0 : 2;       // Y, X, B
      const uint32_t nzr = c == 0 ? nzr0 : c == 1 ? nzr1 : nzr2;
      const uint32_t block_ctx = Uniform(LdS<uint8_t>(oMap + bctx_idx + (uint32_t)ci * bctx_step));
      const uint32_t top = (uint32_t)__builtin_amdgcn_readlane((int)nzr, (int)bx), left = bx ? (uint32_t)__builtin_amdgcn_readlane((int)nzr, (int)(bx - 1)) : 0u;
      const uint32_t pred = bx == 0 ? (by == 0 ? 32u : top) : by == 0 ? left : (top + left + 1) / 2;
      const uint32_t pc = pred > 64 ? 64 : pred;
      const uint32_t nz_ctx = ctx_offset + (pc < 8 ? block_ctx + nctx * pc : block_ctx + nctx * (4 + pc / 2));
      const uint32_t histo = ctx_offset + 37 * nctx + 458 * block_ctx;
      uint32_t nzeros;
      {
        const uint32_t cl = Uniform(LdS<uint8_t>(map_off + nz_ctx));
        const uint32_t cfg = cfg_uniform != 0xFFFFFFFFu ? cfg_uniform : Uniform(LdS<uint32_t>(cfg_off + 4 * cl));
        nzeros = WaveTokenAt(bits, state, wide_off + ((cl << la) << 3), cut_off + ((cl << la) << 1), cfg, la).u;
      }
      if (nzeros + covered > size) { err = kErrNzeros; break; }
      nz_total += nzeros;
      const uint32_t nzm = (nzeros + covered - 1) >> l2;
      if (c == 0) nzr0 = in_cols ? nzm : nzr0; else if (c == 1) nzr1 = in_cols ? nzm : nzr1; else nzr2 = in_cols ? nzm : nzr2;
      if (nzeros == 0) continue;
      // ---- coefficient tokens
      int32_t* const blk = (c == 0 ? cb0 : c == 1 ? cb1 : cb2) + coff;
      const uint16_t* order = pd.orders[ord * 3 + c];
      uint32_t ordv = LdS<uint16_t>(kHwOrdOff + (ord * 3 + (uint32_t)c) * 128 + lane * 2), kbase = 0;
      uint32_t k = covered;
      uint32_t sel = nzeros > size / 16 ? 0u : 1u;
      const uint32_t par = lane & 1;
      // candidates of the first token: the context for "previous was zero" (even lanes) / "was not" (odd lanes) — no count has changed yet
      uint32_t abase, cbase, clus;
      {
        const uint32_t ctx0 = histo + ((uint32_t)LdS<uint8_t>((nzeros + covered - 1) >> l2) + LdS<uint8_t>(64 + (k >> l2))) * 2 + par;
        clus = LdS<uint8_t>(map_off + ctx0);
        abase = wide_off + ((clus << la) << 3); cbase = cut_off + ((clus << la) << 1);
      }
      const uint32_t pmask = (1u << (12 - la)) - 1;
      for (;;) {
        // [A] alias reads, per-lane cluster
        const uint32_t slot = (state & 0xFFF) >> (12 - la), pos = state & pmask, hi = state >> 12, hp = hi + pos;
        const uint2 e = LdS<uint2>(abase + slot * 8);
        const uint32_t cr = LdS<uint16_t>(cbase + slot * 2);
        // [B] candidates of the token after this one: count of non-zeros left if this one is zero (even lanes) / is not (odd lanes)
        const uint32_t nzl = (nzeros - par + covered - 1) >> l2;
        const uint32_t ctxn = histo + ((uint32_t)LdS<uint8_t>(nzl) + LdS<uint8_t>(64 + ((k + 1) >> l2))) * 2 + par;
        const uint32_t clus_n = LdS<uint8_t>(map_off + ctxn);
        // position of this coefficient
        if (__builtin_expect(k >= kbase + 64, 0)) { kbase = k & ~63u; ordv = LdG(order + kbase + lane); __builtin_amdgcn_s_waitcnt(0x0F70); }
        const uint32_t cpos = (uint32_t)__builtin_amdgcn_readlane((int)ordv, (int)(k - kbase));
        // [C] the symbol
        const bool hit = pos >= (cr & 0xFFu);
        const uint32_t cand = hit ? e.y : e.x;
        const uint32_t sw = (uint32_t)__builtin_amdgcn_readlane((int)cand, (int)sel);
        state = (sw & 0xFFFu) * hi + hp + ((sw >> 12) & 0xFFFu);
        int32_t v = (int32_t)sw >> 24;
        if (state < (1u << 16)) { asm volatile("" ::: "memory"); state = (state << 16) | (uint32_t)(bits.buf & 0xFFFFu); bits.buf >>= 16; bits.avail -= 16; bits.Refill(); }
        if (__builtin_expect(v == kWideEscape, 0)) {
          const uint32_t crk = (uint32_t)__builtin_amdgcn_readlane((int)cr, (int)sel);
          uint32_t tok = pos >= (crk & 0xFFu) ? (crk >> 8) : slot;
          uint32_t cfg = cfg_uniform;
          if (cfg == 0xFFFFFFFFu) cfg = Uniform(LdS<uint32_t>(cfg_off + 4 * (uint32_t)__builtin_amdgcn_readlane((int)clus, (int)sel)));
          const uint32_t split_exp = cfg & 0xFF, split = 1u << split_exp;
          if (tok >= split) {
            const uint32_t msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
            const uint32_t nbits = (split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb))) & 31;
            const uint32_t low = tok & ((1u << lsb) - 1);
            tok >>= lsb;
            if ((int)nbits > bits.avail) bits.Refill();
            const uint32_t xb = (uint32_t)(bits.buf & ((1ull << nbits) - 1));
            bits.buf >>= nbits; bits.avail -= (int)nbits;
            const uint32_t hb = (1u << msb) | (tok & ((1u << msb) - 1));
            tok = (((hb << nbits) | xb) << lsb) | low;
            bits.Refill();
          }
          v = UnpackSigned(tok);
          if (tok == 0) v = 0;
        }
        sel = v != 0 ? 1u : 0u;
        if (sel && lane == 0) StG(blk + cpos, v);
        nzeros -= sel;
        k++;
        if (nzeros == 0) break;
        if (k >= size) { err = kErrNzeros; break; }
        clus = clus_n;
        abase = wide_off + ((clus_n << la) << 3); cbase = cut_off + ((clus_n << la) << 1);
      }
    }
  }
  if (lane == 0) {
    if (!err) { if (state != 0x130000u) err = kErrAnsFinalState; else if (bits.BitPos() > limit) err = kErrOverrun; }
    if (err) SetError(f, err);
    else {
      if (f.hf_end_bitpos && f.mod_pass == 0) f.hf_end_bitpos[g] = bits.BitPos();
      if (nz_total) atomicAdd(f.hf_written, nz_total);
    }
  }
}

// =====================================================================================================================
// K_idct: dequant + chroma-from-luma + LLF + inverse transforms.  One 256-thread workgroup per 256x256 group.
// Pass 1 (rows): horizontal 1-D IDCT of every coefficient row, written into the pixel plane as an intermediate;
// 8x8 "special" transforms are completed here.  Pass 2 (columns): vertical 1-D IDCT in place.
// =====================================================================================================================
template <int N> __device__ __forceinline__ void IDct1D(float (&v)[N]) {  // dct-inl.h IDCT1DImpl
  if constexpr (N == 2) { const float a = v[0], b = v[1]; v[0] = a + b; v[1] = a - b; }
  else if constexpr (N > 2) {
    constexpr int H = N / 2;
    float e[H], o[H];
#pragma unroll
    for (int i = 0; i < H; i++) { e[i] = v[2 * i]; o[i] = v[2 * i + 1]; }
    IDct1D<H>(e);
#pragma unroll
    for (int i = H - 1; i > 0; i--) o[i] = o[i] + o[i - 1];
    o[0] = o[0] * 1.41421356237309504880f;
    IDct1D<H>(o);
    constexpr int L = N == 4 ? 2 : N == 8 ? 3 : N == 16 ? 4 : N == 32 ? 5 : 6;
#pragma unroll
    for (int i = 0; i < H; i++) {
      const float mul = d_wc[L][i];
      v[i] = fmaf(mul, o[i], e[i]);
      v[N - 1 - i] = fmaf(-mul, o[i], e[i]);
    }
  }
}

__device__ __forceinline__ float AdjustQuantBias(int32_t q, float bias_c, float bias3) {
  if (q == 0) return 0.0f;
  if (q == 1) return bias_c;
  if (q == -1) return -bias_c;
  const float fq = (float)q;
  return fq - bias3 / fq;
}

struct BlockDequant {
  const int32_t* q[3];      // quantised coefficients of the varblock (Y used for CfL)
  const float* table[3];
  float sdc[3];             // per-channel scaled dequant
  float kx, kb;             // CfL multipliers
  float bias[4];
};

// dequantised coefficient k of channel c (0=X,1=Y,2=B) incl. chroma-from-luma (dec_group.cc DequantLane)
__device__ __forceinline__ float DequantCoef(const BlockDequant& d, int c, uint32_t k) {
  const float y = AdjustQuantBias(d.q[1][k], d.bias[1], d.bias[3]) * (d.table[1][k] * d.sdc[1]);
  if (c == 1) return y;
  const float v = AdjustQuantBias(d.q[c][k], d.bias[c], d.bias[3]) * (d.table[c][k] * d.sdc[c]);
  return fmaf(c == 0 ? d.kx : d.kb, y, v);
}

template <int C> __device__ __forceinline__ void RowPass(const BlockDequant& d, int c, int R, int v, int cy, int cx, const float* llf, size_t llf_stride,
                                                         float* dst /* row start in plane */) {
  float row[C];
#pragma unroll
  for (int u = 0; u < C; u++) {
    const uint32_t k = R >= C ? (uint32_t)(u * R + v) : (uint32_t)(v * C + u);
    row[u] = DequantCoef(d, c, k);
  }
  if (v < cy) {
#pragma unroll
    for (int u = 0; u < C / 8; u++) if (u < cx) row[u] = llf[(size_t)v * llf_stride + u];
  }
  IDct1D<C>(row);
#pragma unroll
  for (int u = 0; u < C; u++) dst[u] = row[u];
}

template <int R> __device__ __forceinline__ void ColPass(float* col0 /* top of column */, size_t stride) {
  float col[R];
#pragma unroll
  for (int v = 0; v < R; v++) col[v] = col0[(size_t)v * stride];
  IDct1D<R>(col);
#pragma unroll
  for (int v = 0; v < R; v++) col0[(size_t)v * stride] = col[v];
}

// 2-D IDCT of a small block held in `sem` (semantic layout [v*C+u]), horizontal first then vertical
template <int R, int C> __device__ __forceinline__ void SmallIdct2D(const float* sem, float* out, size_t stride) {
  float buf[R * C];
#pragma unroll
  for (int v = 0; v < R; v++) {
    float row[C];
#pragma unroll
    for (int u = 0; u < C; u++) row[u] = sem[v * C + u];
    IDct1D<C>(row);
#pragma unroll
    for (int u = 0; u < C; u++) buf[v * C + u] = row[u];
  }
#pragma unroll
  for (int x = 0; x < C; x++) {
    float col[R];
#pragma unroll
    for (int v = 0; v < R; v++) col[v] = buf[v * C + x];
    IDct1D<R>(col);
#pragma unroll
    for (int y = 0; y < R; y++) out[(size_t)y * stride + x] = col[y];
  }
}

// dec_transforms-inl.h: IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4, AFV0-3 on one 8x8 block (coefficients in `cf`, stored layout)
__device__ void SpecialTransform(uint32_t s, const float* cf, float* out, size_t stride) {
  if (s == 1) {  // IDENTITY
    float dcs[4];
    const float b00 = cf[0], b01 = cf[1], b10 = cf[8], b11 = cf[9];
    dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11; dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
    for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
      const float block_dc = dcs[y * 2 + x];
      float residual_sum = 0;
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (ix == 0 && iy == 0) continue; residual_sum += cf[(y + iy * 2) * 8 + x + ix * 2]; }
      const float p11 = block_dc - residual_sum * (1.0f / 16);
      out[(size_t)(4 * y + 1) * stride + 4 * x + 1] = p11;
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) { if (ix == 1 && iy == 1) continue; out[(size_t)(y * 4 + iy) * stride + x * 4 + ix] = cf[(y + iy * 2) * 8 + x + ix * 2] + p11; }
      out[(size_t)(y * 4) * stride + x * 4] = cf[(y + 2) * 8 + x + 2] + p11;
    }
  } else if (s == 2) {  // DCT2X2
    float a[64], b[64];
    for (int i = 0; i < 64; i++) a[i] = cf[i];
    for (int S = 2; S <= 8; S *= 2) {
      const int n = S / 2;
      for (int i = 0; i < 64; i++) b[i] = a[i];
      for (int y = 0; y < n; y++) for (int x = 0; x < n; x++) {
        const float c00 = a[y * 8 + x], c01 = a[y * 8 + n + x], c10 = a[(y + n) * 8 + x], c11 = a[(y + n) * 8 + n + x];
        b[y * 2 * 8 + x * 2] = c00 + c01 + c10 + c11; b[y * 2 * 8 + x * 2 + 1] = c00 + c01 - c10 - c11;
        b[(y * 2 + 1) * 8 + x * 2] = c00 - c01 + c10 - c11; b[(y * 2 + 1) * 8 + x * 2 + 1] = c00 - c01 - c10 + c11;
      }
      for (int i = 0; i < 64; i++) a[i] = b[i];
    }
    for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) out[(size_t)y * stride + x] = a[y * 8 + x];
  } else if (s == 3) {  // DCT4X4
    float dcs[4];
    const float b00 = cf[0], b01 = cf[1], b10 = cf[8], b11 = cf[9];
    dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11; dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
    for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) {
      float sem[16];  // sem[v*4+u] = stored[u*4+v]
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) sem[ix * 4 + iy] = (iy == 0 && ix == 0) ? dcs[y * 2 + x] : cf[(y + iy * 2) * 8 + x + ix * 2];
      SmallIdct2D<4, 4>(sem, out + (size_t)(y * 4) * stride + x * 4, stride);
    }
  } else if (s == 12) {  // DCT4X8: two 4x8 halves stacked vertically
    const float b0 = cf[0], b1 = cf[8];
    const float dcs[2] = {b0 + b1, b0 - b1};
    for (int y = 0; y < 2; y++) {
      float sem[32];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) sem[iy * 8 + ix] = (iy == 0 && ix == 0) ? dcs[y] : cf[(y + iy * 2) * 8 + ix];
      SmallIdct2D<4, 8>(sem, out + (size_t)(y * 4) * stride, stride);
    }
  } else if (s == 13) {  // DCT8X4: two 8x4 halves side by side
    const float b0 = cf[0], b1 = cf[8];
    const float dcs[2] = {b0 + b1, b0 - b1};
    for (int x = 0; x < 2; x++) {
      float sem[32];  // sem[v*4+u] = stored[u*8+v]
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) sem[ix * 4 + iy] = (iy == 0 && ix == 0) ? dcs[x] : cf[(x + iy * 2) * 8 + ix];
      SmallIdct2D<8, 4>(sem, out + x * 4, stride);
    }
  } else {  // s == 14..17, AFV0-3 (AFVTransformToPixels): 4x4 corner in the AFV basis, a 4x4 DCT beside it, a 4x8 DCT in the other half
    const uint32_t afv_x = (s - 14) & 1, afv_y = (s - 14) >> 1;
    const float b00 = cf[0], b01 = cf[1], b10 = cf[8];
    const float dcs[3] = {(b00 + b10 + b01) * 4.0f, (b00 + b10 - b01), b00 - b10};
    {
      float coeff[16];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) coeff[iy * 4 + ix] = (iy == 0 && ix == 0) ? dcs[0] : cf[iy * 2 * 8 + ix * 2];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) {
        const int i = (afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix);
        float px = 0.0f;
        for (int j = 0; j < 16; j++) px = fmaf(coeff[j], d_afv_basis[j][i], px);
        out[(size_t)(iy + afv_y * 4) * stride + afv_x * 4 + ix] = px;
      }
    }
    {
      float sem[16];  // sem[v*4+u] = stored[u*4+v]
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 4; ix++) sem[ix * 4 + iy] = (iy == 0 && ix == 0) ? dcs[1] : cf[iy * 2 * 8 + ix * 2 + 1];
      SmallIdct2D<4, 4>(sem, out + (size_t)(afv_y * 4) * stride + (afv_x == 1 ? 0 : 4), stride);
    }
    {
      float sem[32];
      for (int iy = 0; iy < 4; iy++) for (int ix = 0; ix < 8; ix++) sem[iy * 8 + ix] = (iy == 0 && ix == 0) ? dcs[2] : cf[(1 + iy * 2) * 8 + ix];
      SmallIdct2D<4, 8>(sem, out + (size_t)(afv_y == 1 ? 0 : 4) * stride, stride);
    }
  }
}

// The 8x8 "special" transforms on a block that sits in an LDS tile in stored order, one HALF per lane (two adjacent lanes of a
// wavefront per block and channel).  Half h needs four stored rows — h, h + 2, h + 4, h + 6 (DCT2X2: 2h, 2h + 1, 2h + 4, 2h + 5)
// — plus the DC-mix inputs of rows 0 / 1, and produces rows 4h..4h+3 (DCT8X4: columns 4h..4h+3; AFV: half 0 = the AFV corner and
// the 4x4 DCT beside it from the even rows, half 1 = the 4x8 DCT from the odd rows).  Outputs overwrite the partner's inputs,
// so every lane reads first and all lanes write afterwards (lock-step inside the wavefront + a wave fence).  Same arithmetic,
// same operation order as SpecialTransform; 32 instead of 64 live coefficients per lane.
// SmallIdct2D with the intermediate (row-transformed) block parked at its destination in LDS instead of in registers: the same operations in
// the same order, half the live registers (the half transforms run beside 36 registers of loaded inputs under the tile kernel's 96-VGPR budget)
template <int R, int C, int PITCH> __device__ __forceinline__ void SmallIdct2DLds(const float* sem, float* out) {
#pragma unroll
  for (int v = 0; v < R; v++) {
    float row[C];
#pragma unroll
    for (int u = 0; u < C; u++) row[u] = sem[v * C + u];
    IDct1D<C>(row);
#pragma unroll
    for (int u = 0; u < C; u++) out[v * PITCH + u] = row[u];
  }
  __asm__ volatile("" ::: "memory");     // (keeps the compiler from forwarding the stores above into the loads below: that would be the register version again)
#pragma unroll
  for (int x = 0; x < C; x++) {
    float col[R];
#pragma unroll
    for (int v = 0; v < R; v++) col[v] = out[v * PITCH + x];
    IDct1D<R>(col);
#pragma unroll
    for (int y = 0; y < R; y++) out[y * PITCH + x] = col[y];
  }
}
__device__ __forceinline__ uint32_t SpecialHalfRow(uint32_t s, uint32_t h, int iy) { return s == 2 ? 2 * h + (uint32_t)(iy & 1) + (uint32_t)(iy >> 1) * 4 : h + (uint32_t)iy * 2; }
// DCT2X2: the first two levels only touch the stored 4x4 corner; one lane runs them in place before the halves are read
template <int PITCH> __device__ __forceinline__ void Dct2x2Corner(float* blk) {
  float a[16], b[16];
#pragma unroll
  for (int y = 0; y < 4; y++)
#pragma unroll
    for (int x = 0; x < 4; x++) a[y * 4 + x] = blk[y * PITCH + x];
  {
    const float c00 = a[0], c01 = a[1], c10 = a[4], c11 = a[5];
    a[0] = c00 + c01 + c10 + c11; a[1] = c00 + c01 - c10 - c11; a[4] = c00 - c01 + c10 - c11; a[5] = c00 - c01 - c10 + c11;
  }
#pragma unroll
  for (int y = 0; y < 2; y++)
#pragma unroll
    for (int x = 0; x < 2; x++) {
      const float c00 = a[y * 4 + x], c01 = a[y * 4 + 2 + x], c10 = a[(y + 2) * 4 + x], c11 = a[(y + 2) * 4 + 2 + x];
      b[y * 2 * 4 + x * 2] = c00 + c01 + c10 + c11; b[y * 2 * 4 + x * 2 + 1] = c00 + c01 - c10 - c11;
      b[(y * 2 + 1) * 4 + x * 2] = c00 - c01 + c10 - c11; b[(y * 2 + 1) * 4 + x * 2 + 1] = c00 - c01 - c10 + c11;
    }
#pragma unroll
  for (int y = 0; y < 4; y++)
#pragma unroll
    for (int x = 0; x < 4; x++) blk[y * PITCH + x] = b[y * 4 + x];
}
template <int PITCH> __device__ __forceinline__ void SpecialHalfLoad(uint32_t s, const float* blk, uint32_t h, float (&in)[32], float (&dc)[4]) {
#pragma unroll
  for (int iy = 0; iy < 4; iy++) {
    const float* row = blk + SpecialHalfRow(s, h, iy) * PITCH;
#pragma unroll
    for (int ix = 0; ix < 8; ix++) in[iy * 8 + ix] = row[ix];
  }
  dc[0] = blk[0]; dc[1] = blk[1]; dc[2] = blk[PITCH]; dc[3] = blk[PITCH + 1];   // stored (0,0), (0,1), (1,0), (1,1)
}
template <int PITCH> __device__ __forceinline__ void SpecialHalfStore(uint32_t s, float* blk, uint32_t h, const float (&in)[32], const float (&dc)[4]) {
  if (s == 1) {  // IDENTITY: quadrants (y = h, x = 0, 1); in[iy * 8 + j] = stored (h + 2 iy, j)
    const float b00 = dc[0], b01 = dc[1], b10 = dc[2], b11 = dc[3];
    float dcs[4];
    dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11; dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
#pragma unroll
    for (int x = 0; x < 2; x++) {
      const float block_dc = h == 0 ? dcs[x] : dcs[2 + x];
      float residual_sum = 0;
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++) { if (ix == 0 && iy == 0) continue; residual_sum += in[iy * 8 + x + ix * 2]; }
      const float p11 = block_dc - residual_sum * (1.0f / 16);
      float* out = blk + (h * 4) * PITCH + x * 4;
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++) {
          if (ix == 1 && iy == 1) out[PITCH + 1] = p11;
          else if (ix == 0 && iy == 0) out[0] = in[8 + x + 2] + p11;      // the corner takes the value stored at (1, 1) of the quadrant
          else out[iy * PITCH + ix] = in[iy * 8 + x + ix * 2] + p11;
        }
    }
  } else if (s == 2) {  // DCT2X2, last level: in[] = rows 2h, 2h + 1, 2h + 4, 2h + 5 after Dct2x2Corner; outputs rows 4h .. 4h + 3
#pragma unroll
    for (int yy = 0; yy < 2; yy++)
#pragma unroll
      for (int x = 0; x < 4; x++) {
        const float c00 = in[yy * 8 + x], c01 = in[yy * 8 + 4 + x], c10 = in[(2 + yy) * 8 + x], c11 = in[(2 + yy) * 8 + 4 + x];
        float* out = blk + (h * 4 + yy * 2) * PITCH + x * 2;
        out[0] = c00 + c01 + c10 + c11; out[1] = c00 + c01 - c10 - c11;
        out[PITCH] = c00 - c01 + c10 - c11; out[PITCH + 1] = c00 - c01 - c10 + c11;
      }
  } else if (s == 3) {  // DCT4X4: quadrants (y = h, x = 0, 1)
    const float b00 = dc[0], b01 = dc[1], b10 = dc[2], b11 = dc[3];
    float dcs[4];
    dcs[0] = b00 + b01 + b10 + b11; dcs[1] = b00 + b01 - b10 - b11; dcs[2] = b00 - b01 + b10 - b11; dcs[3] = b00 - b01 - b10 + b11;
#pragma unroll
    for (int x = 0; x < 2; x++) {
      float sem[16];  // sem[v*4+u] = stored[u*4+v]
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 4; ix++) sem[ix * 4 + iy] = (iy == 0 && ix == 0) ? (h == 0 ? dcs[x] : dcs[2 + x]) : in[iy * 8 + x + ix * 2];
      SmallIdct2DLds<4, 4, PITCH>(sem, blk + (h * 4) * PITCH + x * 4);
    }
  } else if (s == 12) {  // DCT4X8: half y = h
    const float b0 = dc[0], b1 = dc[2];
    float sem[32];
#pragma unroll
    for (int iy = 0; iy < 4; iy++)
#pragma unroll
      for (int ix = 0; ix < 8; ix++) sem[iy * 8 + ix] = (iy == 0 && ix == 0) ? (h == 0 ? b0 + b1 : b0 - b1) : in[iy * 8 + ix];
    SmallIdct2DLds<4, 8, PITCH>(sem, blk + (h * 4) * PITCH);
  } else if (s == 13) {  // DCT8X4: half x = h
    const float b0 = dc[0], b1 = dc[2];
    float sem[32];  // sem[v*4+u] = stored[u*8+v]
#pragma unroll
    for (int iy = 0; iy < 4; iy++)
#pragma unroll
      for (int ix = 0; ix < 8; ix++) sem[ix * 4 + iy] = (iy == 0 && ix == 0) ? (h == 0 ? b0 + b1 : b0 - b1) : in[iy * 8 + ix];
    SmallIdct2DLds<8, 4, PITCH>(sem, blk + h * 4);
  } else {  // s == 14..17, AFV0-3: half 0 = even stored rows (AFV corner + the 4x4 DCT beside it), half 1 = odd rows (the 4x8 DCT of the other half)
    const uint32_t afv_x = (s - 14) & 1, afv_y = (s - 14) >> 1;
    const float b00 = dc[0], b01 = dc[1], b10 = dc[2];
    if (h == 0) {
      {
        float coeff[16];
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) coeff[iy * 4 + ix] = (iy == 0 && ix == 0) ? (b00 + b10 + b01) * 4.0f : in[iy * 8 + ix * 2];
        for (int iy = 0; iy < 4; iy++)
          for (int ix = 0; ix < 4; ix++) {
            const int i = (afv_y == 1 ? 3 - iy : iy) * 4 + (afv_x == 1 ? 3 - ix : ix);
            float px = 0.0f;
#pragma unroll
            for (int j = 0; j < 16; j++) px = fmaf(coeff[j], d_afv_basis[j][i], px);
            blk[(iy + afv_y * 4) * PITCH + afv_x * 4 + ix] = px;
          }
      }
      {
        float sem[16];  // sem[v*4+u] = stored[u*4+v]
#pragma unroll
        for (int iy = 0; iy < 4; iy++)
#pragma unroll
          for (int ix = 0; ix < 4; ix++) sem[ix * 4 + iy] = (iy == 0 && ix == 0) ? (b00 + b10 - b01) : in[iy * 8 + ix * 2 + 1];
        SmallIdct2DLds<4, 4, PITCH>(sem, blk + (afv_y * 4) * PITCH + (afv_x == 1 ? 0 : 4));
      }
    } else {
      float sem[32];
#pragma unroll
      for (int iy = 0; iy < 4; iy++)
#pragma unroll
        for (int ix = 0; ix < 8; ix++) sem[iy * 8 + ix] = (iy == 0 && ix == 0) ? b00 - b10 : in[iy * 8 + ix];
      SmallIdct2DLds<4, 8, PITCH>(sem, blk + (afv_y == 1 ? 0 : 4) * PITCH);
    }
  }
}

// One kind of special transform, both phases: the lanes of a wavefront that hold blocks of this kind read their halves, then write them.  The read-before-write
// hazard is between the two halves of ONE block — adjacent lanes with the same kind — so every kind can run as a branch of its own: each loads its inputs in the
// arrangement its transform wants (one generic load ahead of a seven-way branch made the compiler shuffle 32 registers per kind into packed-math pairs, and spill).
// KIND = the strategy (14 stands for AFV0-3: `s` tells which).
template <int PITCH, uint32_t KIND> __device__ __forceinline__ void SpecialHalfRun(uint32_t s, float* blk, uint32_t h) {
  __asm__ volatile("; special transform %0" :: "n"(KIND) : "memory");      // (distinct per kind: keeps the identical leading loads of the branches from being merged ahead of them)
  if (KIND == 2) {
    if (h == 0) Dct2x2Corner<PITCH>(blk);                                  // DCT2X2: the levels below the last one, in place, before either half is read
    WaveSync();
  }
  float in[32], dc[4];
  SpecialHalfLoad<PITCH>(KIND, blk, h, in, dc);
  WaveSync();                                                              // every lane has read its inputs before any lane writes
  SpecialHalfStore<PITCH>(KIND == 14 ? s : KIND, blk, h, in, dc);
}

__device__ __forceinline__ bool IsSpecial(uint32_t s) { return s == 1 || s == 2 || s == 3 || (s >= 12 && s <= 17); }
__device__ __forceinline__ bool IsBig(uint32_t s) { return s >= 21; }
__device__ __forceinline__ uint32_t Log2Cov8(uint32_t n) { return n == 1 ? 0u : n == 2 ? 1u : n == 4 ? 2u : 3u; }   // covered blocks 1, 2, 4, 8   // DCT128x128 ... DCT128x256: larger than a 64x64 tile

__global__ __launch_bounds__(256) void IdctKernel(const FrameDev* __restrict__ frames, int force_generic) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || f.subsampled || ((*f.frame_flags & 1) == 0 && !force_generic) || FrameFailed(f)) return;   // regular frames take IdctTileKernel
  const uint32_t g = blockIdx.x;
  if (g >= f.num_groups) return;
  const uint32_t gx = g % f.xgroups, gy = g / f.xgroups;
  const uint32_t bx0 = gx * 32, by0 = gy * 32;
  const uint32_t gbw = min(32u, f.bw - bx0), gbh = min(32u, f.bh - by0);
  const size_t stride = f.plane_stride;
  // ---- pass 1: rows
  for (uint32_t t = threadIdx.x; t < gbw * gbh * 8; t += blockDim.x) {
    const uint32_t r = t & 7, bi = t >> 3;
    const uint32_t bx = bi % gbw, by = bi / gbw;
    const size_t o = (size_t)(by0 + by) * f.bw + bx0 + bx;
    const uint32_t info = f.blk_info[o];
    if (BI_Ix(info) != 0) continue;                    // rows are owned by the first block column of the varblock
    const uint32_t s = BI_Strategy(info);
    if (IsBig(s)) continue;                            // BigIdctKernel
    const uint32_t iy = BI_Iy(info);
    const int cx = (int)CoveredX(s), cy = (int)CoveredY(s);
    const size_t o_first = o - (size_t)iy * f.bw;      // top-left block of the varblock
    const uint32_t kind = QuantKind(s);
    BlockDequant d;
    const uint32_t coff = f.coef_off[o_first];
    for (int c = 0; c < 3; c++) { d.q[c] = f.coeff[c] + (size_t)g * 65536 + coff; d.table[c] = f.qtable[kind * 3 + c]; }
    const float sd = f.inv_global_scale / (float)BI_HfMul(info);
    d.sdc[0] = sd * f.x_dm; d.sdc[1] = sd; d.sdc[2] = sd * f.b_dm;
    const size_t tile = (size_t)((by0 + by - iy) / 8) * f.cw + (bx0 + bx) / 8;
    d.kx = f.base_x + (float)f.ytox[tile] * f.color_scale;
    d.kb = f.base_b + (float)f.ytob[tile] * f.color_scale;
    for (int i = 0; i < 4; i++) d.bias[i] = f.quant_bias[i];
    const int R = cy * 8, C = cx * 8;
    const int v = (int)(iy * 8 + r);
    if (IsSpecial(s)) {
      if (r != 0) continue;
      for (int c = 0; c < 3; c++) {
        float cf[64];
        for (uint32_t k = 0; k < 64; k++) cf[k] = DequantCoef(d, c, k);
        cf[0] = f.llf[c][o_first];
        SpecialTransform(s, cf, f.plane_a[c] + (size_t)(by0 + by) * 8 * stride + (bx0 + bx) * 8, stride);
      }
      continue;
    }
    for (int c = 0; c < 3; c++) {
      float* dst = f.plane_a[c] + ((size_t)(by0 + by - iy) * 8 + v) * stride + (size_t)(bx0 + bx) * 8;
      const float* llf = f.llf[c] + o_first;
      switch (C) {
        case 8: RowPass<8>(d, c, R, v, cy, cx, llf, f.bw, dst); break;
        case 16: RowPass<16>(d, c, R, v, cy, cx, llf, f.bw, dst); break;
        case 32: RowPass<32>(d, c, R, v, cy, cx, llf, f.bw, dst); break;
        default: RowPass<64>(d, c, R, v, cy, cx, llf, f.bw, dst); break;
      }
    }
  }
  __syncthreads();
  // ---- pass 2: columns
  for (uint32_t t = threadIdx.x; t < gbw * gbh * 8; t += blockDim.x) {
    const uint32_t xx = t & 7, bi = t >> 3;
    const uint32_t bx = bi % gbw, by = bi / gbw;
    const size_t o = (size_t)(by0 + by) * f.bw + bx0 + bx;
    const uint32_t info = f.blk_info[o];
    if (BI_Iy(info) != 0) continue;                    // columns are owned by the first block row of the varblock
    const uint32_t s = BI_Strategy(info);
    if (IsSpecial(s) || IsBig(s)) continue;
    const int R = (int)CoveredY(s) * 8;
    for (int c = 0; c < 3; c++) {
      float* col0 = f.plane_a[c] + (size_t)(by0 + by) * 8 * stride + (size_t)(bx0 + bx) * 8 + xx;
      switch (R) {
        case 8: ColPass<8>(col0, stride); break;
        case 16: ColPass<16>(col0, stride); break;
        case 32: ColPass<32>(col0, stride); break;
        default: ColPass<64>(col0, stride); break;
      }
    }
  }
  // the group's coefficients (all but those of DCT128/256 varblocks, which BigIdctKernel reads after this kernel) have been
  // consumed by pass 1: zero them for the batch's next decode (see IdctTileKernel pass 0)
  const uint32_t count = f.vb_count[g];
  for (uint32_t e = 0; e < count; e++) {
    const uint2 ent = f.vb_list[(size_t)g * 1024 + e];
    const uint32_t s = ent.x & 31;
    if (IsBig(s)) continue;
    const uint32_t n = CoveredX(s) * CoveredY(s) * 64;
    for (uint32_t i = threadIdx.x; i < 3 * n; i += blockDim.x) f.coeff[i / n][(size_t)g * 65536 + ent.y + i % n] = 0;
  }
}

// ---- DCT128 / DCT256 family (strategies 21..26): one workgroup per 256x256 group walks its big varblocks.  The 1-D
// transforms are too long for registers, so each wavefront runs one transform level-parallel in LDS: the recursion of
// IDct1D (even/odd split, adjacent-add of the odd half, half-size transforms, butterflies) is unrolled into log2 N
// split levels and log2 N merge levels that each touch all N elements at once — the same floating-point operations in
// the same order per element as the recursive form, hence bit-identical to it.
constexpr uint32_t kBigWaveFloats = 2 * 256;                  // ping-pong buffers of one wavefront
constexpr uint32_t kBigLlfFloats = 3 * 32 * 32;               // LLF of the current varblock, 3 channels
constexpr uint32_t kBigLds = (kBigLlfFloats + 4 * kBigWaveFloats) * 4;

__device__ __forceinline__ float* WaveIdctLevels(float* a, float* b, int N, int lane) {
  float* src = a; float* dst = b;
  for (int n = N; n >= 4; n >>= 1) {              // split levels
    const int H = n >> 1;
    for (int j = lane; j < N; j += 64) {
      const int base = j & ~(n - 1), r = j & (n - 1);
      float val;
      if (r < H) val = src[base + 2 * r];
      else { const int i = r - H; const float o = src[base + 2 * i + 1]; val = i > 0 ? o + src[base + 2 * i - 1] : o * 1.41421356237309504880f; }
      dst[j] = val;
    }
    WaveSync();
    float* t = src; src = dst; dst = t;
  }
  for (int j = lane; j < N; j += 64) {            // N = 2 leaves
    const float p = src[j & ~1], q = src[j | 1];
    dst[j] = (j & 1) ? p - q : p + q;
  }
  WaveSync();
  { float* t = src; src = dst; dst = t; }
  int L = 2;
  for (int n = 4; n <= N; n <<= 1, L++) {         // merge levels
    const int H = n >> 1;
    for (int j = lane; j < N; j += 64) {
      const int base = j & ~(n - 1), r = j & (n - 1);
      const int i = r < H ? r : n - 1 - r;
      const float e = src[base + i], o = src[base + H + i], mul = d_wc[L][i];
      dst[j] = r < H ? fmaf(mul, o, e) : fmaf(-mul, o, e);
    }
    WaveSync();
    float* t = src; src = dst; dst = t;
  }
  return src;
}

__device__ void FDctDynBig(float* v, int n) {  // n in {1,2,4,8,16,32}; in-register transforms of the small LLF block
  if (n == 2) { float t[2]; for (int i = 0; i < 2; i++) t[i] = v[i]; FDct1D<2>(t); for (int i = 0; i < 2; i++) v[i] = t[i]; }
  else if (n == 4) { float t[4]; for (int i = 0; i < 4; i++) t[i] = v[i]; FDct1D<4>(t); for (int i = 0; i < 4; i++) v[i] = t[i]; }
  else if (n == 8) { float t[8]; for (int i = 0; i < 8; i++) t[i] = v[i]; FDct1D<8>(t); for (int i = 0; i < 8; i++) v[i] = t[i]; }
  else if (n == 16) { float t[16]; for (int i = 0; i < 16; i++) t[i] = v[i]; FDct1D<16>(t); for (int i = 0; i < 16; i++) v[i] = t[i]; }
  else if (n == 32) { float t[32]; for (int i = 0; i < 32; i++) t[i] = v[i]; FDct1D<32>(t); for (int i = 0; i < 32; i++) v[i] = t[i]; }
}
__device__ __forceinline__ int Log2Cov(int n) { return n == 1 ? 0 : n == 2 ? 1 : n == 4 ? 2 : n == 8 ? 3 : n == 16 ? 4 : 5; }

__global__ __launch_bounds__(256) void BigIdctKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || f.subsampled || (*f.frame_flags & 2) == 0 || FrameFailed(f)) return;
  const uint32_t g = blockIdx.x;
  if (g >= f.num_groups) return;
  extern __shared__ __align__(16) float s_big[];
  float* s_llf = s_big;
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* wa = s_big + kBigLlfFloats + wave * kBigWaveFloats;
  float* wbuf = wa + 256;
  const uint32_t gx = g % f.xgroups, gy = g / f.xgroups;
  const size_t stride = f.plane_stride;
  const uint32_t count = f.vb_count[g];
  for (uint32_t e = 0; e < count; e++) {
    const uint2 ent = f.vb_list[(size_t)g * 1024 + e];
    const uint32_t s = ent.x & 31;
    if (!IsBig(s)) continue;
    const uint32_t bx = gx * 32 + ((ent.x >> 16) & 31), by = gy * 32 + ((ent.x >> 21) & 31);
    const uint32_t hf_mul = BI_HfMul(f.blk_info[(size_t)by * f.bw + bx]);
    const int cx = (int)CoveredX(s), cy = (int)CoveredY(s), R = cy * 8, C = cx * 8;
    const size_t o_first = (size_t)by * f.bw + bx;
    // ---- LLF: scaled forward DCT of the varblock's LF samples (columns, then rows — LlfSigmaKernel's order)
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 3u * cx; t += blockDim.x) {
      const uint32_t c = t / cx, xx = t % cx;
      float col[32];
      for (int yy = 0; yy < cy; yy++) col[yy] = f.lf_tmp[c][o_first + (size_t)yy * f.bw + xx];
      FDctDynBig(col, cy);
      const float sr = 1.0f / (float)cy;
      for (int v = 0; v < cy; v++) s_llf[c * 1024 + v * 32 + xx] = col[v] * sr;
    }
    __syncthreads();
    for (uint32_t t = threadIdx.x; t < 3u * cy; t += blockDim.x) {
      const uint32_t c = t / cy, v = t % cy;
      float row[32];
      for (int u = 0; u < cx; u++) row[u] = s_llf[c * 1024 + v * 32 + u];
      FDctDynBig(row, cx);
      const float sc = 1.0f / (float)cx;
      const int lx = Log2Cov(cx), ly = Log2Cov(cy);
      for (int u = 0; u < cx; u++) s_llf[c * 1024 + v * 32 + u] = ((row[u] * sc) * d_resample[ly][v]) * d_resample[lx][u];
    }
    __syncthreads();
    // ---- pass 1: rows — dequant + chroma-from-luma + LLF substitution, horizontal transform, row written to the plane
    BlockDequant d;
    const uint32_t kind = QuantKind(s);
    for (int c = 0; c < 3; c++) { d.q[c] = f.coeff[c] + (size_t)g * 65536 + ent.y; d.table[c] = f.qtable[kind * 3 + c]; }
    const float sd = f.inv_global_scale / (float)hf_mul;
    d.sdc[0] = sd * f.x_dm; d.sdc[1] = sd; d.sdc[2] = sd * f.b_dm;
    for (int i = 0; i < 4; i++) d.bias[i] = f.quant_bias[i];
    {  // chroma-from-luma factors of the tile holding the varblock's first block (dec_group.cc)
      const size_t t0 = (size_t)(by / 8) * f.cw + bx / 8;
      d.kx = f.base_x + (float)f.ytox[t0] * f.color_scale;
      d.kb = f.base_b + (float)f.ytob[t0] * f.color_scale;
    }
    for (uint32_t task = wave; task < 3u * R; task += 4) {
      const int c = (int)(task / R), v = (int)(task % R);
      for (int u = (int)lane; u < C; u += 64) {
        const uint32_t k = R >= C ? (uint32_t)(u * R + v) : (uint32_t)(v * C + u);
        wa[u] = (v < cy && u < cx) ? s_llf[c * 1024 + v * 32 + u] : DequantCoef(d, c, k);
      }
      WaveSync();
      const float* res = WaveIdctLevels(wa, wbuf, C, (int)lane);
      float* dst = f.plane_a[c] + ((size_t)by * 8 + v) * stride + (size_t)bx * 8;
      for (int u = (int)lane; u < C; u += 64) dst[u] = res[u];
      WaveSync();
    }
    __threadfence();
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < 3u * R * C; i += blockDim.x) f.coeff[i / (R * C)][(size_t)g * 65536 + ent.y + i % (R * C)] = 0;   // consumed: zero for the next decode
    // ---- pass 2: columns in place
    for (uint32_t task = wave; task < 3u * C; task += 4) {
      const int c = (int)(task / C), xx = (int)(task % C);
      float* col0 = f.plane_a[c] + (size_t)by * 8 * stride + (size_t)bx * 8 + xx;
      for (int v = (int)lane; v < R; v += 64) wa[v] = col0[(size_t)v * stride];
      WaveSync();
      const float* res = WaveIdctLevels(wa, wbuf, R, (int)lane);
      for (int v = (int)lane; v < R; v += 64) col0[(size_t)v * stride] = res[v];
      WaveSync();
    }
  }
}

// ---- chroma-subsampled frames (YCbCr JPEG transcodes; dec_group.cc with !Is444()): 8x8 DCT only, no chroma-from-luma, every channel on
// its own block grid.  One thread per (block, channel) of a 256x256 group; a channel takes part in a block only where the block starts
// one of its cells, and writes its 8x8 pixels at the cell's position of the packed (top-left) subsampled plane; the chroma planes are
// brought to full resolution afterwards (ChromaUpsampleKernel).  Same arithmetic as the 8x8 path of IdctKernel (rows, then columns).
__global__ __launch_bounds__(256) void IdctSubsampledKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || !f.subsampled || FrameFailed(f)) return;
  const uint32_t g = blockIdx.x;
  if (g >= f.num_groups) return;
  const uint32_t gx = g % f.xgroups, gy = g / f.xgroups;
  const uint32_t bx0 = gx * 32, by0 = gy * 32;
  const uint32_t gbw = min(32u, f.bw - bx0), gbh = min(32u, f.bh - by0);
  const size_t stride = f.plane_stride;
  for (uint32_t t = threadIdx.x; t < gbw * gbh * 3; t += blockDim.x) {
    const uint32_t c = t / (gbw * gbh), bi = t - c * gbw * gbh;
    const uint32_t X = bx0 + bi % gbw, Y = by0 + bi / gbw;
    if ((X & ((1u << f.hs[c]) - 1u)) | (Y & ((1u << f.vs[c]) - 1u))) continue;
    const size_t o = (size_t)Y * f.bw + X;
    const uint32_t info = LdG(f.blk_info + o);
    if (BI_Strategy(info) != 0) continue;                      // (rejected by the LF stage)
    int32_t* q = f.coeff[c] + (size_t)g * 65536 + LdG(f.coef_off + o);
    const float* table = f.qtable[c];                          // kind 0 (DCT8), channel c
    const float sd = f.inv_global_scale / (float)BI_HfMul(info);
    const float sdc = c == 0 ? sd * f.x_dm : c == 2 ? sd * f.b_dm : sd;
    const float bias_c = f.quant_bias[c], bias3 = f.quant_bias[3];
    float sem[64];
    for (uint32_t k = 0; k < 64; k++) {
      const int32_t v = q[k];
      if (v) q[k] = 0;                                         // consumed: the planes stay clean for the next decode
      sem[(k & 7) * 8 + (k >> 3)] = AdjustQuantBias(v, bias_c, bias3) * (table[k] * sdc);   // stored layout is the transpose of (v, u)
    }
    const uint32_t sx = X >> f.hs[c], sy = Y >> f.vs[c];
    sem[0] = LdG(f.lf[c] + (size_t)sy * f.bw + sx);            // the lowest frequency of an 8x8 DCT is the LF sample
    SmallIdct2D<8, 8>(sem, f.plane_a[c] + (size_t)sy * 8 * stride + (size_t)sx * 8, stride);
  }
}

// The same on tiles (round 6): a 256-thread workgroup takes 8 x 4 blocks of ONE channel (64 x 32 samples of its packed plane) through LDS — the thread-per-block form above reads
// its 64 coefficients 256 bytes apart from its neighbours' and writes 8-sample rows with the plane's stride between lanes (30 ms per 256 4K frames, 272 bytes of scratch).
//  pass 0: coalesced read of the tile's 2048 quantised coefficients (consecutive lanes = consecutive coefficients of a block), dequantisation, the LF sample in slot 0,
//          into the block's 8 x 8 cell in LDS at the coefficient's (v, u) position; zeros back where something was read
//  pass 1: one row per thread (IDct1D<8>)      pass 2: one column per thread      pass 3: coalesced write of the 64 x 32 samples
// Arithmetic and operation order: SmallIdct2D<8, 8> (rows, then columns) on the values IdctSubsampledKernel computes.
constexpr int kSubTileBx = 8, kSubTileBy = 4, kSubCell = 73;            // (cells of 8 rows x 9 floats + 1: rows and columns of neighbouring blocks fall on different banks)
__global__ __launch_bounds__(256) void IdctSubsampledTileKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || !f.subsampled || FrameFailed(f)) return;
  const uint32_t c = blockIdx.y, hs = f.hs[c], vs = f.vs[c];
  const uint32_t cbw = f.bw >> hs, cbh = f.bh >> vs;                     // the channel's own block grid (bw, bh are multiples of the largest cell)
  const uint32_t tiles_x = (cbw + kSubTileBx - 1) / kSubTileBx, tiles_y = (cbh + kSubTileBy - 1) / kSubTileBy;
  if (blockIdx.x >= tiles_x * tiles_y) return;
  const uint32_t sx0 = (blockIdx.x % tiles_x) * kSubTileBx, sy0 = (blockIdx.x / tiles_x) * kSubTileBy;
  __shared__ float s_px[kSubTileBx * kSubTileBy * kSubCell];
  __shared__ int32_t* s_q[kSubTileBx * kSubTileBy];
  __shared__ float s_mul[kSubTileBx * kSubTileBy], s_lf[kSubTileBx * kSubTileBy];
  const uint32_t t = threadIdx.x;
  if (t < kSubTileBx * kSubTileBy) {
    const uint32_t sx = sx0 + t % kSubTileBx, sy = sy0 + t / kSubTileBx;
    int32_t* q = nullptr; float mul = 0.f, lf = 0.f;
    if (sx < cbw && sy < cbh) {
      const uint32_t X = sx << hs, Y = sy << vs;
      const size_t o = (size_t)Y * f.bw + X;
      const uint32_t info = LdG(f.blk_info + o);
      if (BI_Strategy(info) == 0) {                                      // (anything else was rejected by the LF stage)
        const uint32_t g = (Y / 32) * f.xgroups + X / 32;
        q = f.coeff[c] + (size_t)g * 65536 + LdG(f.coef_off + o);
        const float sd = f.inv_global_scale / (float)BI_HfMul(info);
        mul = c == 0 ? sd * f.x_dm : c == 2 ? sd * f.b_dm : sd;
        lf = LdG(f.lf[c] + (size_t)sy * f.bw + sx);                      // the lowest frequency of an 8x8 DCT is the LF sample
      }
    }
    s_q[t] = q; s_mul[t] = mul; s_lf[t] = lf;
  }
  __syncthreads();
  const float* table = f.qtable[c];                                      // kind 0 (DCT8), channel c
  const float bias_c = f.quant_bias[c], bias3 = f.quant_bias[3];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t i = t + 256u * (uint32_t)j, b = i >> 6, k = i & 63;
    int32_t* q = s_q[b];
    float v = 0.f;
    if (q) {
      const int32_t qv = LdG(q + k);
      if (qv) StG(q + k, 0);                                             // consumed: the planes stay clean for the next decode
      v = k == 0 ? s_lf[b] : AdjustQuantBias(qv, bias_c, bias3) * (LdG(table + k) * s_mul[b]);
    }
    s_px[b * kSubCell + (k & 7) * 9 + (k >> 3)] = v;                     // stored layout is the transpose of (v, u)
  }
  __syncthreads();
  {
    const uint32_t b = t >> 3, r = t & 7;
    float* row = s_px + b * kSubCell + r * 9;
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; u++) v[u] = row[u];
    IDct1D<8>(v);
#pragma unroll
    for (int u = 0; u < 8; u++) row[u] = v[u];
  }
  __syncthreads();
  {
    const uint32_t b = t >> 3, x = t & 7;
    float* col = s_px + b * kSubCell + x;
    float v[8];
#pragma unroll
    for (int y = 0; y < 8; y++) v[y] = col[y * 9];
    IDct1D<8>(v);
#pragma unroll
    for (int y = 0; y < 8; y++) col[y * 9] = v[y];
  }
  __syncthreads();
  const size_t stride = f.plane_stride;
  float* out = f.plane_a[c];
#pragma unroll
  for (int j = 0; j < 8; j++) {
    const uint32_t i = t + 256u * (uint32_t)j, px = i & 63, py = i >> 6;  // sample (px, py) of the 64 x 32 tile
    const uint32_t b = (py >> 3) * kSubTileBx + (px >> 3);
    if (!s_q[b]) continue;
    out[(size_t)(sy0 * 8 + py) * stride + sx0 * 8 + px] = s_px[b * kSubCell + (py & 7) * 9 + (px & 7)];
  }
}

// ---- fast path: one 256-thread workgroup per 64x64-pixel tile (8x8 blocks), all three channels staged in LDS ----------
// Requires every varblock to lie inside one tile (true for naturally aligned blocks, i.e. everything encoders emit);
// frames violating that are flagged by the LF stage and use IdctKernel above.  Same arithmetic, same operation order.
//  pass 0: coalesced read of the quantised coefficients (consecutive lanes = consecutive coefficients), dequant + CfL,
//          scatter into the LDS tile at the coefficient's (vertical, horizontal) frequency position
//  pass 1: LLF substitution + horizontal 1-D IDCT per row (in LDS)     pass 2: vertical 1-D IDCT per column (in LDS)
//  pass 3: coalesced write of the finished 64x64 tile to the three planes
// TB = tile side in 8x8 blocks: 8 (64x64 pixels, any varblock up to 64x64) or 4 (32x32 pixels: frames whose varblocks are at
// most 32x32 — a quarter of the LDS per workgroup, four times the workgroups in flight to cover barrier waits)
template <int TB> struct TileGeom { static constexpr int kPx = TB * 8, kPitch = TB * 8 + 1, kPlane = TB * 8 * (TB * 8 + 1); };

template <int C> __device__ __forceinline__ void TileRowPass(float* row0 /* LDS row start; the LLF samples are already in place */) {
  float row[C];
#pragma unroll
  for (int u = 0; u < C; u++) row[u] = row0[u];
  IDct1D<C>(row);
#pragma unroll
  for (int u = 0; u < C; u++) row0[u] = row[u];
}

template <int R, int PITCH> __device__ __forceinline__ void TileColPass(float* col0) {
  float col[R];
#pragma unroll
  for (int v = 0; v < R; v++) col[v] = col0[v * PITCH];
  IDct1D<R>(col);
#pragma unroll
  for (int v = 0; v < R; v++) col0[v * PITCH] = col[v];
}

// SPECIAL = the variant for frames that contain the 8x8 "special" transforms (IDENTITY, DCT2X2, DCT4X4, DCT4X8, DCT8X4):
// their 64-coefficient register blocks cost 30 VGPRs that the plain variant does not have to carry (the LF stage flags
// the frames; which variants a batch needs is known after its first decode).
#ifdef JXL_IDCT_NUM_VGPR   // (A/B knob: a register budget between what the waves-per-SIMD steps of __launch_bounds__ give — 128, 96; hipcc 7.2 keeps 110 VGPRs for <4, true> under 104: not honoured)
#define JXL_IDCT_VGPR_ATTR __attribute__((amdgpu_num_vgpr(JXL_IDCT_NUM_VGPR)))
#else
#define JXL_IDCT_VGPR_ATTR
#endif
template <int TB, bool SPECIAL> __global__ __launch_bounds__(TB == 8 ? 256 : JXL_IDCT_T4, TB == 8 ? 2 : JXL_IDCT_MINW) JXL_IDCT_VGPR_ATTR void IdctTileKernel(const FrameDev* __restrict__ frames, int tiles_x, int force_generic) {
  constexpr int kTilePitch = TileGeom<TB>::kPitch, kTilePlane = TileGeom<TB>::kPlane, kNB = TB * TB;
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || f.subsampled || (*f.frame_flags & 1) != 0 || (force_generic & 3)) return;
  if (((*f.frame_flags & 4) != 0) != (TB == 8)) return;     // frames with a varblock that no 32x32 tile contains take the 64x64 tiles
  if (((*f.frame_flags & 8) != 0) != SPECIAL) return;
  const uint32_t tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  if (tx * TB >= f.bw || ty * TB >= f.bh) return;
  extern __shared__ __align__(16) float s_tile[];   // 3 * kTilePlane floats
  __shared__ uint32_t s_info[kNB];
  __shared__ uint32_t s_coff[kNB];
  __shared__ float s_llf[3 * kNB];
  // bias[3] / |q| for |q| < 128: the smart dequantisation bias needs one correctly rounded division per non-trivial
  // coefficient (12 per task, ~10 instructions each); the same quotients come out of a table filled by 128 divisions per tile
  __shared__ float s_bias_q[128];
  if (threadIdx.x < 128) s_bias_q[threadIdx.x] = threadIdx.x ? f.quant_bias[3] / (float)threadIdx.x : 0.0f;
  const uint32_t bx0 = tx * TB, by0 = ty * TB;
  const uint32_t tbw = min((uint32_t)TB, f.bw - bx0), tbh = min((uint32_t)TB, f.bh - by0);
  const uint32_t g = (by0 / 32) * f.xgroups + bx0 / 32;
  __shared__ uint32_t s_failed;
  if (threadIdx.x == blockDim.x - 1) s_failed = __hip_atomic_load(f.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (FrameFailed: read beside the block infos, looked at behind the barrier they need anyway)
  if (threadIdx.x < kNB) {
    const uint32_t bx = threadIdx.x % TB, by = threadIdx.x / TB;
    uint32_t info = 0xFFFFFFFFu, coff = 0;
    if (bx < tbw && by < tbh) {
      const size_t o = (size_t)(by0 + by) * f.bw + bx0 + bx;
      info = LdG(f.blk_info + o);
      coff = LdG(f.coef_off + o);                   // offset of the covering varblock (stored for all of its blocks)
      // the LLF plane holds one sample per 8x8 block: sample (iy, ix) of the covering varblock's low-frequency corner
      for (int c = 0; c < 3; c++) s_llf[c * kNB + threadIdx.x] = LdG(f.llf[c] + o);
    }
    s_info[threadIdx.x] = info; s_coff[threadIdx.x] = coff;
  }
  __syncthreads();
  if (s_failed) return;
  if (IsBig(BI_Strategy(s_info[0]))) return;   // aligned DCT128/256 varblocks cover whole tiles: BigIdctKernel owns them
  // ---- row / column task lists sorted by transform length.  A tile mixes 8-, 16-, 32- and 64-point rows; taken in
  // raster order every wavefront would run all four unrolled transforms one after the other (divergence), sorted by
  // length a wavefront runs one.  Classes 0..3 = 8 << class points, class 4 = the 8x8 "special" transforms.
  __shared__ uint32_t s_cnt[16];                 // [0..4] row counts, [5..8] column counts, then running cursors
  __shared__ uint16_t s_rtask[kNB * 8], s_ctask[kNB * 8];
  __shared__ uint32_t s_next;                    // next unassigned row task of pass 1
  if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
  if (threadIdx.x == 0) s_next = 0;
  __syncthreads();
  constexpr int kQ = (kNB * 8 + (TB == 8 ? 256 : JXL_IDCT_T4) - 1) / (TB == 8 ? 256 : JXL_IDCT_T4);   // candidate row/column tasks per thread
  uint32_t my_rclass[kQ], my_cclass[kQ];
  for (int q = 0; q < kQ; q++) {
    const uint32_t tt = threadIdx.x + q * blockDim.x;
    my_rclass[q] = my_cclass[q] = 0xFFu;
    if (tt >= (uint32_t)kNB * 8) continue;
    const uint32_t info = s_info[tt >> 3];
    if (info == 0xFFFFFFFFu) continue;
    const uint32_t st = BI_Strategy(info);
    if (SPECIAL && IsSpecial(st)) { if ((tt & 7) == 0) my_rclass[q] = 4; continue; }   // one task per special block (rows pass only)
    if (BI_Ix(info) == 0) my_rclass[q] = Log2Cov8(CoveredX(st));
    if (BI_Iy(info) == 0) my_cclass[q] = Log2Cov8(CoveredY(st));
  }
  for (int q = 0; q < kQ; q++) {
    if (my_rclass[q] != 0xFFu) atomicAdd(&s_cnt[my_rclass[q]], 1u);
    if (my_cclass[q] != 0xFFu) atomicAdd(&s_cnt[5 + my_cclass[q]], 1u);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t a = 0;
    for (int k = 0; k < 5; k++) { s_cnt[9 + k] = a; a += s_cnt[k]; }          // row cursors (exclusive prefix)
    // column cursors share the tail of the array: classes 0..3 only; class index 5..8 counts
  }
  __shared__ uint32_t s_ccur[4];
  if (threadIdx.x == 0) { uint32_t a = 0; for (int k = 0; k < 4; k++) { s_ccur[k] = a; a += s_cnt[5 + k]; } }
  __syncthreads();
  // (the same for every lane: kept in scalar registers — eleven VGPRs that the special transforms' 36 loaded inputs need under the 96-register budget)
  auto uni = [](uint32_t v) -> uint32_t { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); };
  const uint32_t r_begin[6] = {uni(s_cnt[9]), uni(s_cnt[10]), uni(s_cnt[11]), uni(s_cnt[12]), uni(s_cnt[13]), uni(s_cnt[13] + s_cnt[4])};
  const uint32_t c_begin[5] = {uni(s_ccur[0]), uni(s_ccur[1]), uni(s_ccur[2]), uni(s_ccur[3]), uni(s_ccur[3] + s_cnt[8])};
  __syncthreads();
  for (int q = 0; q < kQ; q++) {
    const uint32_t tt = threadIdx.x + q * blockDim.x;
    if (my_rclass[q] != 0xFFu) s_rtask[atomicAdd(&s_cnt[9 + my_rclass[q]], 1u)] = (uint16_t)tt;
    if (my_cclass[q] != 0xFFu) s_ctask[atomicAdd(&s_ccur[my_cclass[q]], 1u)] = (uint16_t)tt;
  }
  int32_t* cq[3] = {f.coeff[0] + (size_t)g * 65536, f.coeff[1] + (size_t)g * 65536, f.coeff[2] + (size_t)g * 65536};
  const float bias0 = f.quant_bias[0], bias1 = f.quant_bias[1], bias2 = f.quant_bias[2], bias3 = f.quant_bias[3];
  // ---- pass 0: stage dequantised coefficients.  Task = (block of the tile, four of its 64 coefficient slots): 16-byte
  // loads, consecutive lanes read consecutive 16-byte chunks.  The tile is exactly one chroma-from-luma tile.
  const size_t cfl_i = (size_t)(by0 / 8) * f.cw + bx0 / 8;   // chroma-from-luma factors are per 64x64 pixels
  const float kx = f.base_x + (float)LdG(f.ytox + cfl_i) * f.color_scale;
  const float kb = f.base_b + (float)LdG(f.ytob + cfl_i) * f.color_scale;
  auto adjust = [&](int32_t q, float bias_c) -> float {   // AdjustQuantBias with the quotient from the table (x / -y == -(x / y) exactly)
    if (q == 0) return 0.0f;
    const uint32_t aq = (uint32_t)(q < 0 ? -q : q);
    if (aq == 1) return q < 0 ? -bias_c : bias_c;
    const float fq = (float)q;
    float d;
    if (aq < 128) d = copysignf(s_bias_q[aq], fq); else d = bias3 / fq;
    return fq - d;
  };
  struct Pass0Task { uint32_t info, bi, k0; int4 qy, qx, qb; float4 ty, tx, tb; };
  auto p0_load = [&](uint32_t t, Pass0Task& p) {   // issues the six 16-byte loads of task t (if it exists)
    p.info = 0xFFFFFFFFu;
    if (t >= (uint32_t)kNB * 16) return;
    p.bi = t >> 4;
    p.info = s_info[p.bi];
    if (p.info == 0xFFFFFFFFu) return;
    const uint32_t s = BI_Strategy(p.info), ix = BI_Ix(p.info), iy = BI_Iy(p.info);
    p.k0 = (iy * CoveredX(s) + ix) * 64 + (t & 15) * 4;      // this block's share of the varblock's coefficients
    const uint32_t kind = QuantKind(s), base = s_coff[p.bi] + p.k0;
    if (force_generic & 4) {      // experiment (JXL_HIP_IDCT_NOCOEF, wrong pixels): what the stage would cost without the dense coefficient read
      p.qy = p.qx = p.qb = make_int4(0, 0, 0, (int)(t & 1));
    } else {
    p.qy = LdG(reinterpret_cast<const int4*>(cq[1] + base));
    p.qx = LdG(reinterpret_cast<const int4*>(cq[0] + base));
    p.qb = LdG(reinterpret_cast<const int4*>(cq[2] + base));
    }
    // The HF stage only writes non-zero coefficients, so the planes must be zero again before the batch's next decode: every
    // kernel that consumes a block's coefficients for good puts zeros back where it found something else (a few MB per frame
    // instead of a 100 MB memset).
    {
      const int4 z = make_int4(0, 0, 0, 0);
      if (p.qy.x | p.qy.y | p.qy.z | p.qy.w) StG(reinterpret_cast<int4*>(cq[1] + base), z);
      if (p.qx.x | p.qx.y | p.qx.z | p.qx.w) StG(reinterpret_cast<int4*>(cq[0] + base), z);
      if (p.qb.x | p.qb.y | p.qb.z | p.qb.w) StG(reinterpret_cast<int4*>(cq[2] + base), z);
    }
    p.ty = LdG(reinterpret_cast<const float4*>(f.qtable[kind * 3 + 1] + p.k0));
    p.tx = LdG(reinterpret_cast<const float4*>(f.qtable[kind * 3 + 0] + p.k0));
    p.tb = LdG(reinterpret_cast<const float4*>(f.qtable[kind * 3 + 2] + p.k0));
  };
  auto p0_store = [&](const Pass0Task& p) {        // dequant + chroma-from-luma + scatter to (v, u)
    if (p.info == 0xFFFFFFFFu) return;
    const uint32_t info = p.info, bi = p.bi, k0 = p.k0;
    const uint32_t s = BI_Strategy(info), ix = BI_Ix(info), iy = BI_Iy(info);
    const uint32_t cx = CoveredX(s), cy = CoveredY(s);
    const uint32_t R = cy * 8, C = cx * 8;
    const uint32_t lr = 3 + Log2Cov8(cy), lc = 3 + Log2Cov8(cx);
    const float sd = f.inv_global_scale / (float)BI_HfMul(info);
    const float sdx = sd * f.x_dm, sdb = sd * f.b_dm;
    const int32_t qy[4] = {p.qy.x, p.qy.y, p.qy.z, p.qy.w}, qx[4] = {p.qx.x, p.qx.y, p.qx.z, p.qx.w}, qb[4] = {p.qb.x, p.qb.y, p.qb.z, p.qb.w};
    const float wy[4] = {p.ty.x, p.ty.y, p.ty.z, p.ty.w}, wx[4] = {p.tx.x, p.tx.y, p.tx.z, p.tx.w}, wbl[4] = {p.tb.x, p.tb.y, p.tb.z, p.tb.w};
    const uint32_t vbx = (bi % TB) - ix, vby = (bi / TB) - iy;   // varblock origin inside the tile (blocks)
    const bool special = SPECIAL && IsSpecial(s);
    // the four coefficients k0 .. k0+3 (k0 a multiple of 4) are neighbours along one axis of the (v, u) grid
    uint32_t v0, u0, step;
    if (special) { v0 = k0 >> 3; u0 = k0 & 7; step = 1; }                        // kept in stored order for SpecialTransform
    else if (R >= C) { v0 = k0 & (R - 1); u0 = k0 >> lr; step = kTilePitch; }    // (R, C are powers of two)
    else { v0 = k0 >> lc; u0 = k0 & (C - 1); step = 1; }
    const uint32_t lo0 = (vby * 8 + v0) * kTilePitch + vbx * 8 + u0;
#pragma unroll
    for (int e = 0; e < 4; e++) {
      const float ydq = adjust(qy[e], bias1) * (wy[e] * sd);
      const float xv = adjust(qx[e], bias0) * (wx[e] * sdx);
      const float bv = adjust(qb[e], bias2) * (wbl[e] * sdb);
      const uint32_t lo = lo0 + (uint32_t)e * step;
      s_tile[lo] = fmaf(kx, ydq, xv);
      s_tile[kTilePlane + lo] = ydq;
      s_tile[2 * kTilePlane + lo] = fmaf(kb, ydq, bv);
    }
  };
  // two tasks per thread in flight: the loads of the second are issued before the first is dequantised
  for (uint32_t t = threadIdx.x; t < (uint32_t)kNB * 16; t += 2 * blockDim.x) {
    Pass0Task pa, pb;
    p0_load(t, pa);
    p0_load(t + blockDim.x, pb);
    p0_store(pa);
    p0_store(pb);
  }
  __syncthreads();
  // ---- the lowest frequencies come from the LF image: every block of the tile puts its LLF sample where its varblock's
  // coefficient (iy, ix) sits (special 8x8 transforms: their single one at the block origin)
  for (uint32_t t = threadIdx.x; t < 3u * kNB; t += blockDim.x) {
    const uint32_t c = t / kNB, bi = t - c * kNB, info = s_info[bi];
    if (info == 0xFFFFFFFFu) continue;
    const uint32_t ix = BI_Ix(info), iy = BI_Iy(info), bx = bi % TB, by = bi / TB;
    s_tile[c * kTilePlane + ((by - iy) * 8 + iy) * kTilePitch + (bx - ix) * 8 + ix] = s_llf[t];
  }
  __syncthreads();
  // ---- pass 1: rows.  The regular row tasks (sorted by transform length, x 3 channels) are handed out in chunks of 64
  // from a counter in LDS; the 8x8 special transforms — one lane per (block, channel), ~10 times the work of a row — are
  // taken first by the last wavefront while the others already pull rows (run after the rows they doubled this phase).
  {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6;
    auto block_of = [&](uint32_t tt, uint32_t c, uint32_t& s, uint32_t& iy, size_t& o_first) -> float* {
      const uint32_t bi = tt >> 3, info = s_info[bi];
      s = BI_Strategy(info); iy = BI_Iy(info);
      const uint32_t bx = bi % TB, by = bi / TB;
      o_first = (size_t)(by0 + by - iy) * f.bw + bx0 + bx;
      return s_tile + c * kTilePlane + ((by - iy) * 8) * kTilePitch + bx * 8;
    };
    if constexpr (SPECIAL) {
      const uint32_t n4 = r_begin[5] - r_begin[4];
      if (n4 && wave == nwaves - 1) {
        for (uint32_t t0 = 0; t0 < n4 * 6; t0 += 64) {     // (block, channel, half) per lane; both halves in adjacent lanes
          uint32_t ln = lane;
          __asm__ volatile("" : "+v"(ln));                   // (opaque: what derives from the lane is computed in here, not ahead of the loop and then spilled)
          const uint32_t t = t0 + ln;
          const bool live = t < n4 * 6;
          const uint32_t task = live ? t >> 1 : 0, h = t & 1;
          const uint32_t c = (task >= n4 ? 1u : 0u) + (task >= 2 * n4 ? 1u : 0u), tt = s_rtask[r_begin[4] + (task - c * n4)];
          uint32_t s, iy; size_t o_first;
          float* blk0 = block_of(tt, c, s, iy, o_first);
          if (live) {
            if (s == 3) SpecialHalfRun<kTilePitch, 3>(s, blk0, h);
            else if (s == 12) SpecialHalfRun<kTilePitch, 12>(s, blk0, h);
            else if (s == 13) SpecialHalfRun<kTilePitch, 13>(s, blk0, h);
            else if (s == 1) SpecialHalfRun<kTilePitch, 1>(s, blk0, h);
            else if (s == 2) SpecialHalfRun<kTilePitch, 2>(s, blk0, h);
            else SpecialHalfRun<kTilePitch, 14>(s, blk0, h);
          }
        }
      }
    }
    const uint32_t b1 = r_begin[1] - r_begin[0], b2 = r_begin[2] - r_begin[0], b3 = r_begin[3] - r_begin[0], nreg = r_begin[4] - r_begin[0];
    while (true) {
      uint32_t base = 0;
      if (lane == 0) base = atomicAdd(&s_next, 64u);
      base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
      if (base >= nreg * 3) break;
      const uint32_t t = base + lane;
      if (t >= nreg * 3) continue;
      const uint32_t c = t / nreg, idx = t - c * nreg, tt = s_rtask[r_begin[0] + idx];
      const uint32_t cls = idx < b1 ? 0 : idx < b2 ? 1 : idx < b3 ? 2 : 3;
      uint32_t s, iy; size_t o_first;
      float* blk0 = block_of(tt, c, s, iy, o_first);
      float* row0 = blk0 + (iy * 8 + (tt & 7)) * kTilePitch;
      if (cls == 0) TileRowPass<8>(row0);
      else if (cls == 1) TileRowPass<16>(row0);
      else if (cls == 2) TileRowPass<32>(row0);
      else if (TB == 8) TileRowPass<64>(row0);
    }
  }
  __syncthreads();
  // ---- pass 2: columns (one list over all transform lengths x 3 channels: no partly filled pass per length)
  {
    const uint32_t b1 = c_begin[1] - c_begin[0], b2 = c_begin[2] - c_begin[0], b3 = c_begin[3] - c_begin[0], ncol = c_begin[4] - c_begin[0];
    for (uint32_t t = threadIdx.x; t < ncol * 3; t += blockDim.x) {
      const uint32_t c = t / ncol, idx = t - c * ncol, tt = s_ctask[c_begin[0] + idx];
      const uint32_t cls = idx < b1 ? 0 : idx < b2 ? 1 : idx < b3 ? 2 : 3;
      const uint32_t xx = tt & 7, bi = tt >> 3;
      const uint32_t bx = bi % TB, by = bi / TB;
      float* col0 = s_tile + c * kTilePlane + (by * 8) * kTilePitch + bx * 8 + xx;
      if (cls == 0) TileColPass<8, kTilePitch>(col0);
      else if (cls == 1) TileColPass<16, kTilePitch>(col0);
      else if (cls == 2) TileColPass<32, kTilePitch>(col0);
      else if (TB == 8) TileColPass<64, kTilePitch>(col0);
    }
  }
  __syncthreads();
  // ---- pass 3: coalesced write of the tile into the planes
  const uint32_t tw = tbw * 8, th = tbh * 8;
  for (uint32_t c = 0; c < 3; c++) {
    float* dst = f.plane_a[c] + (size_t)(by0 * 8) * f.plane_stride + bx0 * 8;
    const float* src = s_tile + c * kTilePlane;
    for (uint32_t i = threadIdx.x; i < (uint32_t)(TB * 2) * th; i += blockDim.x) {   // 16 bytes per store (rows are 32-byte aligned)
      const uint32_t y = i / (TB * 2), x = (i % (TB * 2)) * 4;
      const float* sp = src + y * kTilePitch + x;
      if (x < tw) StG(reinterpret_cast<float4*>(dst + (size_t)y * f.plane_stride + x), make_float4(sp[0], sp[1], sp[2], sp[3]));
    }
  }
}

// =====================================================================================================================
// K_gab / K_epf: loop restoration filters on the (w x h) image with mirrored borders
// =====================================================================================================================
__device__ __forceinline__ int MirrorD(int x, int size) {
  while (x < 0 || x >= size) x = x < 0 ? -x - 1 : 2 * size - 1 - x;
  return x;
}
__device__ __forceinline__ int FilterStagesBefore(const FrameDev& f, int stage, bool gab_folded = false) {  // stage: 0 gab, 1 epf0, 2 epf1, 3 epf2; gab_folded: gaborish runs inside the first EPF pass (no plane of its own)
  int n = 0;
  if (stage > 0 && f.gab && !gab_folded) n++;
  if (stage > 1 && f.epf_iters >= 3) n++;
  if (stage > 2 && f.epf_iters >= 1) n++;
  if (stage > 3 && f.epf_iters >= 2) n++;
  return n;
}
__device__ __forceinline__ bool FilterStageActive(const FrameDev& f, int stage) {
  return stage == 0 ? f.gab != 0 : stage == 1 ? f.epf_iters >= 3 : stage == 2 ? f.epf_iters >= 1 : f.epf_iters >= 2;
}

// Frames with the common restoration setting (gaborish + one EPF pass, XYB colour) take the fused tile kernel
// FusedGabEpf1OutKernel; everything else runs the stage-by-stage kernels.
__device__ __forceinline__ bool FusedEligible(const FrameDev& f, int unfused) { return !unfused && f.gab && f.epf_iters == 1 && f.color_mode <= 1 && f.upsampling == 1 && f.post_mode == 0; }

// frames whose last EPF pass writes the pixels itself (EpfTileKernel; fuse_out bit 0 clear: a test stops the tail between the stages)
__device__ __forceinline__ bool EpfWritesOutput(const FrameDev& f, int unfused, int fuse_out) {
  return (fuse_out & 1) && !FusedEligible(f, unfused) && f.epf_iters >= 1 && f.upsampling == 1 && f.post_mode == 0;
}
// ... and whose first EPF pass applies gaborish to its own tile while loading it (fuse_out bit 1; nothing but the EPF kernels looks at such a frame's planes afterwards,
// so the plane the separate gaborish pass would have filled is never missed)
__device__ __forceinline__ bool GabFolded(const FrameDev& f, int unfused, int fuse_out) { return (fuse_out & 2) && f.gab && EpfWritesOutput(f, unfused, fuse_out); }

__global__ void GaborishKernel(const FrameDev* __restrict__ frames, int unfused, int fuse_out) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || !f.gab || FusedEligible(f, unfused) || GabFolded(f, unfused, fuse_out)) return;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  const int w = (int)f.width, h = (int)f.height;
  if (x >= w || y >= h) return;
  const size_t stride = f.plane_stride;
  const int yt = MirrorD(y - 1, h), yb = MirrorD(y + 1, h), xl = MirrorD(x - 1, w), xr = MirrorD(x + 1, w);
  for (int c = 0; c < 3; c++) {
    const float* src = f.plane_a[c];
    const float* t = src + (size_t)yt * stride; const float* m = src + (size_t)y * stride; const float* b = src + (size_t)yb * stride;
    const float sum0 = m[x];
    const float sum1 = (m[xl] + m[xr]) + (t[x] + b[x]);
    const float sum2 = (t[xl] + t[xr]) + (b[xl] + b[xr]);
    f.plane_b[c][(size_t)y * stride + x] = fmaf(sum2, f.gab_w[c * 3 + 2], fmaf(sum1, f.gab_w[c * 3 + 1], sum0 * f.gab_w[c * 3 + 0]));
  }
}


// =====================================================================================================================
// K_out: XYB -> linear -> sRGB -> clamp/scale/round -> interleaved caller layout (stage_xyb/from_linear/write)
// =====================================================================================================================
__device__ __forceinline__ float LinearToSrgb(float v) {
  const float x = fabsf(v);
  const float lin = x * 12.92f;
  const float s = sqrtf(x);
  float yp = 7.352629620e-1f, yq = 2.424867759e-2f;
  yp = fmaf(yp, s, 1.474205315f); yq = fmaf(yq, s, 9.258482155e-1f);
  yp = fmaf(yp, s, 3.903842876e-1f); yq = fmaf(yq, s, 1.340816930f);
  yp = fmaf(yp, s, 5.287254571e-3f); yq = fmaf(yq, s, 3.036675394e-1f);
  yp = fmaf(yp, s, -5.135152395e-4f); yq = fmaf(yq, s, 1.004519624e-2f);
  const float poly = yp / yq;
  return copysignf(x > 0.0031308f ? poly : lin, v);
}

// base/fast_math-inl.h FastLog2f / FastPow2f / FastPowf and the transfer functions built on them (stage_from_linear.cc OpGamma, TF_709)
__device__ __forceinline__ float FastPowfDev(float base, float exponent) {
  const int32_t x_bits = __float_as_int(base);
  const int32_t exp_shifted = (x_bits - 0x3f2aaaab) >> 23;
  const float t = __int_as_float(x_bits - (int32_t)((uint32_t)exp_shifted << 23)) - 1.0f;
  float yp = fmaf(7.4245873327820566E-01f, t, 1.4287160470083755E+00f); yp = fmaf(yp, t, -1.8503833400518310E-06f);
  float yq = fmaf(1.7409343003366853E-01f, t, 1.0096718572241148E+00f); yq = fmaf(yq, t, 9.9032814277590719E-01f);
  const float x = (yp / yq + (float)exp_shifted) * exponent;
  const float floorx = floorf(x);
  const float exp = __int_as_float((int32_t)((uint32_t)((int32_t)floorx + 127) << 23));
  const float frac = x - floorx;
  float num = frac + 1.01749063e+01f;
  num = fmaf(num, frac, 4.88687798e+01f);
  num = fmaf(num, frac, 9.85506591e+01f);
  num = num * exp;
  float den = fmaf(frac, 2.10242958e-01f, -2.22328856e-02f);
  den = fmaf(den, frac, -1.94414990e+01f);
  den = fmaf(den, frac, 9.85506633e+01f);
  return num / den;
}
__device__ __forceinline__ float GammaFromLinear(float v, float inverse_gamma) { return v <= 1e-5f ? 0.0f : FastPowfDev(v, inverse_gamma); }
__device__ __forceinline__ float Rec709FromLinear(float v) { return v <= 0.018f ? 4.5f * v : fmaf(1.099f, FastPowfDev(v, 0.45f), -0.099f); }

__device__ __forceinline__ uint16_t FloatToHalfBits(float fv) {
  const uint32_t x = __float_as_uint(fv);
  const uint32_t sign = (x >> 16) & 0x8000;
  const int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t mant = x & 0x7FFFFF;
  if (((x >> 23) & 0xFF) == 0xFF) return (uint16_t)(sign | 0x7C00 | (mant ? 0x200 : 0));
  if (exp >= 31) return (uint16_t)(sign | 0x7C00);
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    mant |= 0x800000;
    const int shift = 14 - exp;
    uint32_t m = mant >> shift;
    const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (m & 1))) m++;
    return (uint16_t)(sign | m);
  }
  const uint32_t m = mant >> 13, rem = mant & 0x1FFF;
  uint32_t r = (uint32_t)(exp << 10) | m;
  if (rem > 0x1000 || (rem == 0x1000 && (m & 1))) r++;
  return (uint16_t)(sign | r);
}

__device__ __forceinline__ void StoreSample(const FrameDev& f, uint8_t* p, float v) {
  if (f.out_type == 0) {
    p[0] = (uint8_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, v)) * f.out_int_mul);
  } else if (f.out_type == 1) {
    const uint32_t u = (uint32_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, v)) * f.out_int_mul);
    if (f.out_big_endian) { p[0] = (uint8_t)(u >> 8); p[1] = (uint8_t)u; } else { p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); }
  } else if (f.out_type == 2) {
    const uint32_t u = __float_as_uint(v);
    if (f.out_big_endian) { p[0] = (uint8_t)(u >> 24); p[1] = (uint8_t)(u >> 16); p[2] = (uint8_t)(u >> 8); p[3] = (uint8_t)u; }
    else { p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); p[2] = (uint8_t)(u >> 16); p[3] = (uint8_t)(u >> 24); }
  } else {
    const uint32_t u = FloatToHalfBits(v);
    if (f.out_big_endian) { p[0] = (uint8_t)(u >> 8); p[1] = (uint8_t)u; } else { p[0] = (uint8_t)u; p[1] = (uint8_t)(u >> 8); }
  }
}

// Position of image sample (x, y) in the output buffer: the header's orientation (1..8, EXIF numbering as in
// codestream_header.rs JxlOrientation) is applied by the write stage — 2 flip-h, 3 rotate 180, 4 flip-v, 5 transpose,
// 6 rotate 90 cw, 7 anti-transpose, 8 rotate 90 ccw; out_stride already refers to the oriented width.
__device__ __forceinline__ uint8_t* OutPixelPtr(const FrameDev& f, int x, int y, uint32_t bps) {
  const int w = (int)f.img_w, h = (int)f.img_h;
  int ox = x, oy = y;
  switch (f.out_orient) {
    case 2: ox = w - 1 - x; break;
    case 3: ox = w - 1 - x; oy = h - 1 - y; break;
    case 4: oy = h - 1 - y; break;
    case 5: ox = y; oy = x; break;
    case 6: ox = h - 1 - y; oy = x; break;
    case 7: ox = h - 1 - y; oy = w - 1 - x; break;
    case 8: ox = y; oy = w - 1 - x; break;
    default: break;
  }
  return f.out + (size_t)oy * f.out_stride + (size_t)ox * f.out_channels * bps;
}

// a whole dword of output samples (JXL_PACKED_STORE_NT: with the non-temporal hint — nothing on the device reads decoded pixels again)
__device__ __forceinline__ void StoreOut32(uint32_t* p, uint32_t v) {
#ifdef JXL_PACKED_STORE_NT
  __builtin_nontemporal_store(v, p);
#else
  *p = v;
#endif
}

__device__ __forceinline__ void StorePixel(const FrameDev& f, int x, int y, float r, float g, float b, float a) {
  const uint32_t bps = f.out_type == 0 ? 1 : f.out_type == 2 ? 4 : 2;
  uint8_t* p = OutPixelPtr(f, x, y, bps);
  const uint32_t nc = f.out_channels;
  if (nc <= 2) {
    StoreSample(f, p, f.is_gray ? r : g);  // gray images carry the same value in all channels; otherwise take G
    if (nc == 2) StoreSample(f, p + bps, a);
  } else {
    StoreSample(f, p, r); StoreSample(f, p + bps, g); StoreSample(f, p + 2 * bps, b);
    if (nc == 4) StoreSample(f, p + 3 * bps, a);
  }
}

// Non-separable 2x / 4x / 8x upsampling of the restored planes (stage_upsampling.cc; same definition and accumulation order
// as oracle/render.h UpsamplePlane): one thread per output sample and channel, 25 taps, result clamped to the window's range.
// Channel 3 = the alpha extra channel (int samples scaled to float first).
__global__ void UpsampleKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || f.upsampling == 1 || f.post_mode) return;
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y * blockDim.y + threadIdx.y;
  if (ox >= (int)f.img_w || oy >= (int)f.img_h) return;
  const int up = (int)f.upsampling, N = up / 2;
  const int w = (int)f.width, h = (int)f.height;
  const int x = ox / up, sx = ox % up, y = oy / up, sy = oy % up;
  const int ky = sy < N ? sy : up - 1 - sy, kx = sx < N ? sx : up - 1 - sx;
  const bool fy = sy >= N, fx = sx >= N;
  const bool src_is_a = (FilterStagesBefore(f, 4) & 1) == 0;
  const int nch = f.alpha_plane ? 4 : 3;
  for (int c = 0; c < nch; c++) {
    const float* src = c == 3 ? nullptr : (src_is_a ? f.plane_a[c] : f.plane_b[c]);
    float sum = 0.0f, mn = 0.0f, mx = 0.0f;
    for (int iy = 0; iy < 5; iy++) {
      const int yy = MirrorD(y + iy - 2, h);
      const int mi = 5 * ky + (fy ? 4 - iy : iy);
      for (int ix = 0; ix < 5; ix++) {
        const int xx = MirrorD(x + ix - 2, w);
        const float v = c == 3 ? (float)f.alpha_plane[(size_t)yy * w + xx] * f.alpha_factor : src[(size_t)yy * f.plane_stride + xx];
        const int mj = 5 * kx + (fx ? 4 - ix : ix);
        const int lo = mi < mj ? mi : mj, hi = mi < mj ? mj : mi;
        const float k = f.up_weights[5 * N * lo - lo * (lo - 1) / 2 + hi - lo];
        sum = fmaf(k, v, sum);
        if (iy == 0 && ix == 0) { mn = v; mx = v; } else { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
      }
    }
    f.up_plane[c][(size_t)oy * f.img_w + ox] = sum < mn ? mn : (sum > mx ? mx : sum);
  }
}

__device__ __forceinline__ uint32_t XcdContiguous(uint32_t bid, uint32_t nwg);
// XYB (or YCbCr / RGB) sample of image position (x, y) -> colour transform -> transfer function -> the caller's layout: the tail of OutputKernel, shared with the last
// EPF pass of the tiled filter path (EpfTileKernel), which hands its results over in registers instead of through a plane
__device__ __forceinline__ void ColorAndStore(const FrameDev& f, int x, int y, float X, float Y, float B, float A) {
  float r, g, b;
  if (f.color_mode <= 1 || f.color_mode >= 4) {
    const float gr = (Y + X) - f.neg_bias_cbrt[0];
    const float gg = (Y - X) - f.neg_bias_cbrt[1];
    const float gb = B - f.neg_bias_cbrt[2];
    const float mr = fmaf(gr * gr, gr, f.neg_bias[0]);
    const float mg = fmaf(gg * gg, gg, f.neg_bias[1]);
    const float mb = fmaf(gb * gb, gb, f.neg_bias[2]);
    r = fmaf(f.opsin_inv[2], mb, fmaf(f.opsin_inv[1], mg, f.opsin_inv[0] * mr));
    g = fmaf(f.opsin_inv[5], mb, fmaf(f.opsin_inv[4], mg, f.opsin_inv[3] * mr));
    b = fmaf(f.opsin_inv[8], mb, fmaf(f.opsin_inv[7], mg, f.opsin_inv[6] * mr));
    if (f.color_mode == 0) { r = LinearToSrgb(r); g = LinearToSrgb(g); b = LinearToSrgb(b); }
    else if (f.color_mode == 4) { r = GammaFromLinear(r, f.inverse_gamma); g = GammaFromLinear(g, f.inverse_gamma); b = GammaFromLinear(b, f.inverse_gamma); }
    else if (f.color_mode == 5) { r = Rec709FromLinear(r); g = Rec709FromLinear(g); b = Rec709FromLinear(b); }
    else if (f.color_mode == 6) { r = PqFromLinear(r, f.hdr_par[0]); g = PqFromLinear(g, f.hdr_par[0]); b = PqFromLinear(b, f.hdr_par[0]); }
    else if (f.color_mode == 7) {
      HlgInverseOotf(f.hdr_par, r, g, b, [](float x, float e) { return FastPowfDev(x, e); });
      r = HlgFromLinear(r); g = HlgFromLinear(g); b = HlgFromLinear(b);
    }
  } else if (f.color_mode == 2) {
    const float c128 = 128.0f / 255, crcr = 1.402f, cgcb = -0.114f * 1.772f / 0.587f, cgcr = -0.299f * 1.402f / 0.587f, cbcb = 1.772f;
    const float yb = Y + c128;
    r = fmaf(crcr, B, yb);
    g = fmaf(cgcr, B, fmaf(cgcb, X, yb));
    b = fmaf(cbcb, X, yb);
  } else { r = X; g = Y; b = B; }
  if (f.is_gray) r = g;
  StorePixel(f, x, y, r, g, b, A);
}
// Sample (x, y) of channel c of a chroma-subsampled frame at full resolution (stage_chroma_upsampling.cc: horizontal, then vertical, each with the (1/4, 3/4) kernel — out[2x] = 0.25 in[x-1]
// + 0.75 in[x], out[2x+1] = 0.25 in[x+1] + 0.75 in[x] —, neighbours clamped at the channel's own edges; the channel sits packed in the top-left corner of its plane).  The arithmetic and
// its order are ChromaUpsampleKernel's (kernels_features.hip: the frame tail of images with features): the vertical step works on horizontally upsampled rows, exactly as two stages would.
__device__ __forceinline__ float SubsampledAt(const FrameDev& f, const float* __restrict__ plane, int c, uint32_t x, uint32_t y) {
  const uint32_t hs = f.hs[c], vs = f.vs[c];
  if (!(hs | vs)) return plane[(size_t)y * f.plane_stride + x];
  const uint32_t cw = (f.width + (1u << hs) - 1) >> hs, ch = (f.height + (1u << vs) - 1) >> vs;
  const uint32_t sx = x >> hs, sy = y >> vs;
  auto hval = [&](uint32_t row) -> float {
    const float* in = plane + (size_t)row * f.plane_stride;
    if (!hs) return in[x];
    const float mid = in[sx] * 0.75f;
    const uint32_t nb = (x & 1) ? min(sx + 1, cw - 1) : (sx ? sx - 1 : 0);
    return fmaf(0.25f, in[nb], mid);
  };
  if (!vs) return hval(sy);
  const float mid = hval(sy) * 0.75f;
  const uint32_t nb = (y & 1) ? min(sy + 1, ch - 1) : (sy ? sy - 1 : 0);
  return fmaf(0.25f, hval(nb), mid);
}
__global__ void OutputKernel(const FrameDev* __restrict__ frames, int unfused, int fuse_out) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || f.post_mode || FusedEligible(f, unfused) || EpfWritesOutput(f, unfused, fuse_out)) return;
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= (int)f.img_w || y >= (int)f.img_h) return;
  float X, Y, B, A = 1.0f;
  if (f.subsampled) {
    // JPEG transcodes with subsampled chroma (no restoration filters, no upsampling — the host rejects those combinations): chroma upsampling, YCbCr -> RGB and the write in one pass
    X = SubsampledAt(f, f.plane_a[0], 0, (uint32_t)x, (uint32_t)y);
    Y = SubsampledAt(f, f.plane_a[1], 1, (uint32_t)x, (uint32_t)y);
    B = SubsampledAt(f, f.plane_a[2], 2, (uint32_t)x, (uint32_t)y);
    if (f.alpha_plane) A = (float)f.alpha_plane[(size_t)y * f.width + x] * f.alpha_factor;
  } else if (f.upsampling > 1) {
    const size_t o = (size_t)y * f.img_w + x;
    X = f.up_plane[0][o]; Y = f.up_plane[1][o]; B = f.up_plane[2][o];
    if (f.alpha_plane) A = f.up_plane[3][o];
  } else {
    const bool src_is_a = (FilterStagesBefore(f, 4) & 1) == 0;
    const size_t o = (size_t)y * f.plane_stride + x;
    X = (src_is_a ? f.plane_a[0] : f.plane_b[0])[o];
    Y = (src_is_a ? f.plane_a[1] : f.plane_b[1])[o];
    B = (src_is_a ? f.plane_a[2] : f.plane_b[2])[o];
    if (f.alpha_plane) A = (float)f.alpha_plane[(size_t)y * f.width + x] * f.alpha_factor;
  }
  ColorAndStore(f, x, y, X, Y, B, A);
}

// =====================================================================================================================
// EPF pass on 32x32 pixel tiles out of LDS (frames the fused gaborish + EPF 1 kernel does not take: two or three EPF iterations, other transfer functions,
// no gaborish ...).  The stage-by-stage kernel this replaces read every tap from global memory with mirrored coordinates worked out per tap — pass 0 is 12 taps x 3
// channels x 5-sample plus shapes x 2 = 360 loads per pixel — and spilled 90 registers.  Here a workgroup loads the tile with a halo of 3 / 2 / 1 samples (the
// stage's reach: taps + plus shape) once, mirrored at the image border like every stage of libjxl's pipeline mirrors its own input, and every thread works four
// pixels out of LDS with the arithmetic and operation order of stage_epf.cc (= EpfKernel): results are bit-identical.  The last active pass hands its pixels
// straight to the colour transform and the write stage (no plane in between) unless something else follows (upsampling, the frame tail of complex images).
// =====================================================================================================================
constexpr int kEtT = 32;                                   // tile edge
template <int PASS> struct EpfTileGeom { static constexpr int kHalo = PASS == 0 ? 3 : PASS == 1 ? 2 : 1, kR = kEtT + 2 * kHalo, kP = kR + 1; };
// One pixel of EPF pass PASS out of an LDS tile: t0 = the pixel's sample of channel 0, `plane` floats between the channels, row pitch P (stage_epf.cc's arithmetic and operation order)
template <int PASS> __device__ __forceinline__ void EpfPixel(const FrameDev& f, const float* t0, int plane, int P, int x, int y, float cs0, float cs1, float cs2, float (&out)[3]) {
  constexpr int ntaps = PASS == 0 ? 12 : 4;
  constexpr int taps0[12][2] = {{0, -2}, {-1, -1}, {0, -1}, {1, -1}, {-2, 0}, {-1, 0}, {1, 0}, {2, 0}, {-1, 1}, {0, 1}, {1, 1}, {0, 2}};
  constexpr int taps1[4][2] = {{0, -1}, {-1, 0}, {1, 0}, {0, 1}};
  constexpr int plus[5][2] = {{0, 0}, {0, -1}, {-1, 0}, {1, 0}, {0, 1}};
  auto px = [&](int c, int dx, int dy) -> float { return t0[c * plane + dy * P + dx]; };
  const float is = f.inv_sigma[(size_t)(y / 8) * f.bw + x / 8];
  if (is < -3.90524291751269967465540850526868f) { out[0] = px(0, 0, 0); out[1] = px(1, 0, 0); out[2] = px(2, 0, 0); return; }
  const bool border = (x % 8 == 0) || (x % 8 == 7) || (y % 8 == 0) || (y % 8 == 7);
  const float vmul = is * (border ? f.epf_bsm[PASS] : f.epf_sm[PASS]);
  float wsum = 1.0f;
  float acc[3] = {px(0, 0, 0), px(1, 0, 0), px(2, 0, 0)};
#pragma unroll
  for (int t = 0; t < ntaps; t++) {
    const int dx = PASS == 0 ? taps0[t][0] : taps1[t][0], dy = PASS == 0 ? taps0[t][1] : taps1[t][1];
    float sad = 0.f;
    if (PASS == 2) {
      sad = fmaf(fabsf(px(0, dx, dy) - px(0, 0, 0)), cs0, sad);
      sad = fmaf(fabsf(px(1, dx, dy) - px(1, 0, 0)), cs1, sad);
      sad = fmaf(fabsf(px(2, dx, dy) - px(2, 0, 0)), cs2, sad);
    } else {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        float sacc = 0.f;
#pragma unroll
        for (int k = 0; k < 5; k++) sacc += fabsf(px(c, dx + plus[k][0], dy + plus[k][1]) - px(c, plus[k][0], plus[k][1]));
        sad = fmaf(sacc, c == 0 ? cs0 : c == 1 ? cs1 : cs2, sad);
      }
    }
    const float wgt = fmaxf(0.0f, fmaf(sad, vmul, 1.0f));
    wsum += wgt;
#pragma unroll
    for (int c = 0; c < 3; c++) acc[c] = fmaf(wgt, px(c, dx, dy), acc[c]);
  }
  const float inv = 1.0f / wsum;
#pragma unroll
  for (int c = 0; c < 3; c++) out[c] = acc[c] * inv;
}
// frames whose passes 1 and 2 run in ONE kernel (EpfTile12Kernel; fuse_out bit 2): the last pass writes the pixels, and pass 1 is not the pass that carries a folded gaborish
__device__ __forceinline__ bool EpfPasses12Fused(const FrameDev& f, int unfused, int fuse_out) {
  return (fuse_out & 4) && f.epf_iters >= 2 && EpfWritesOutput(f, unfused, fuse_out) && !(GabFolded(f, unfused, fuse_out) && f.epf_iters < 3);
}
template <int PASS> __global__ __launch_bounds__(256) void EpfTileKernel(const FrameDev* __restrict__ frames, int unfused, int tiles_x, int fuse_out) {
  const FrameDev& f = frames[blockIdx.z];
  constexpr int stage = PASS + 1;
  if (f.is_modular || !FilterStageActive(f, stage) || FusedEligible(f, unfused)) return;
  if (PASS >= 1 && EpfPasses12Fused(f, unfused, fuse_out)) return;          // EpfTile12Kernel
  const int w = (int)f.width, h = (int)f.height;
  const uint32_t tile = XcdContiguous(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % (uint32_t)tiles_x) * kEtT, y0 = (int)(tile / (uint32_t)tiles_x) * kEtT;
  if (x0 >= w || y0 >= h) return;
  constexpr int H = EpfTileGeom<PASS>::kHalo, R = EpfTileGeom<PASS>::kR, P = EpfTileGeom<PASS>::kP;
  __shared__ float s_t[3 * R * P];
  const bool folded = GabFolded(f, unfused, fuse_out);
  const bool src_is_a = (FilterStagesBefore(f, stage, folded) & 1) == 0;
  const size_t stride = f.plane_stride;
  const float* src[3]; float* dst[3];
#pragma unroll
  for (int c = 0; c < 3; c++) { src[c] = src_is_a ? f.plane_a[c] : f.plane_b[c]; dst[c] = src_is_a ? f.plane_b[c] : f.plane_a[c]; }
  // the frame's first EPF pass of a frame with folded gaborish: the tile is loaded with one more sample of halo, gaborish (GaborishKernel's arithmetic) fills the EPF tile
  // for the positions inside the image, and the positions outside it take the value of their mirror image — what this pass would have read from a gaborish plane
  constexpr bool kCanFold = PASS < 2;                      // (pass 2 is never the first one)
  constexpr int RG = R + 2, PG = RG + 1;
  __shared__ float s_raw[kCanFold ? 3 * RG * PG : 1];
  const bool fold_here = kCanFold && folded && (PASS == 0 || f.epf_iters < 3);
  if (fold_here) {
    for (int i = threadIdx.x; i < RG * RG; i += 256) {
      const int ry = i / RG, rx = i - ry * RG;
      const size_t o = (size_t)MirrorD(y0 + ry - H - 1, h) * stride + MirrorD(x0 + rx - H - 1, w);
#pragma unroll
      for (int c = 0; c < 3; c++) s_raw[(c * RG + ry) * PG + rx] = LdG(f.plane_a[c] + o);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * R; i += 256) {
      const int ly = i / R, lx = i - ly * R;
      const int Y = y0 + ly - H, X = x0 + lx - H;
      if (Y < 0 || Y >= h || X < 0 || X >= w) continue;
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float* t = s_raw + (c * RG + ly) * PG + lx; const float* m = t + PG; const float* b = m + PG;      // rows Y - 1, Y, Y + 1 from column X - 1
        const float sum0 = m[1];
        const float sum1 = (m[0] + m[2]) + (t[1] + b[1]);
        const float sum2 = (t[0] + t[2]) + (b[0] + b[2]);
        s_t[(c * R + ly) * P + lx] = fmaf(sum2, f.gab_w[c * 3 + 2], fmaf(sum1, f.gab_w[c * 3 + 1], sum0 * f.gab_w[c * 3 + 0]));
      }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < R * R; i += 256) {
      const int ly = i / R, lx = i - ly * R;
      const int Y = y0 + ly - H, X = x0 + lx - H;
      if (Y >= 0 && Y < h && X >= 0 && X < w) continue;
      const int my = MirrorD(Y, h) - y0 + H, mx = MirrorD(X, w) - x0 + H;
      if (my < 0 || my >= R || mx < 0 || mx >= R) continue;                     // (farther out than any pixel of the image reaches)
#pragma unroll
      for (int c = 0; c < 3; c++) s_t[(c * R + ly) * P + lx] = s_t[(c * R + my) * P + mx];
    }
  } else {
    for (int i = threadIdx.x; i < R * R; i += 256) {
      const int ly = i / R, lx = i - ly * R;
      const size_t o = (size_t)MirrorD(y0 + ly - H, h) * stride + MirrorD(x0 + lx - H, w);
#pragma unroll
      for (int c = 0; c < 3; c++) s_t[(c * R + ly) * P + lx] = LdG(src[c] + o);
    }
  }
  __syncthreads();
  const bool last = PASS == 2 || (PASS == 1 && f.epf_iters == 1);
  const bool write_out = last && EpfWritesOutput(f, unfused, fuse_out);
  const float cs0 = f.epf_channel_scale[0], cs1 = f.epf_channel_scale[1], cs2 = f.epf_channel_scale[2];
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    const int lx = threadIdx.x & 31, ly = (threadIdx.x >> 5) + 8 * q;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) continue;
    float out[3];
    EpfPixel<PASS>(f, s_t + (ly + H) * P + (lx + H), R * P, P, x, y, cs0, cs1, cs2, out);
    if (write_out) {
      const float A = f.alpha_plane ? (float)f.alpha_plane[(size_t)y * f.width + x] * f.alpha_factor : 1.0f;
      ColorAndStore(f, x, y, out[0], out[1], out[2], A);
    } else {
      const size_t o = (size_t)y * stride + x;
#pragma unroll
      for (int c = 0; c < 3; c++) dst[c][o] = out[c];
    }
  }
}

// EPF passes 1 and 2 in one kernel (frames with two or three EPF iterations whose last pass writes the pixels): the tile is loaded with the halo of both passes (3), pass 1 is
// computed for the 34x34 positions pass 2 reads — inside the image; positions outside it take the value of their mirror image, what pass 2 would have read from pass 1's plane —
// into a second LDS tile, and pass 2 runs out of that.  One pass over the planes less; per pixel the arithmetic of the two separate passes (EpfPixel).
__global__ __launch_bounds__(256) void EpfTile12Kernel(const FrameDev* __restrict__ frames, int unfused, int tiles_x, int fuse_out) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || FusedEligible(f, unfused) || !EpfPasses12Fused(f, unfused, fuse_out)) return;
  const int w = (int)f.width, h = (int)f.height;
  const uint32_t tile = XcdContiguous(blockIdx.x, gridDim.x);
  const int x0 = (int)(tile % (uint32_t)tiles_x) * kEtT, y0 = (int)(tile / (uint32_t)tiles_x) * kEtT;
  if (x0 >= w || y0 >= h) return;
  constexpr int R0 = kEtT + 6, P0 = R0 + 1, R1 = kEtT + 2, P1 = R1 + 1;
  __shared__ float s_in[3 * R0 * P0];
  __shared__ float s_p1[3 * R1 * P1];
  const bool src_is_a = (FilterStagesBefore(f, 2, GabFolded(f, unfused, fuse_out)) & 1) == 0;
  const size_t stride = f.plane_stride;
  for (int i = threadIdx.x; i < R0 * R0; i += 256) {
    const int ly = i / R0, lx = i - ly * R0;
    const size_t o = (size_t)MirrorD(y0 + ly - 3, h) * stride + MirrorD(x0 + lx - 3, w);
#pragma unroll
    for (int c = 0; c < 3; c++) s_in[(c * R0 + ly) * P0 + lx] = LdG((src_is_a ? f.plane_a[c] : f.plane_b[c]) + o);
  }
  __syncthreads();
  const float cs0 = f.epf_channel_scale[0], cs1 = f.epf_channel_scale[1], cs2 = f.epf_channel_scale[2];
#pragma unroll 1
  for (int i = threadIdx.x; i < R1 * R1; i += 256) {
    const int ly = i / R1, lx = i - ly * R1;
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    if (y < 0 || y >= h || x < 0 || x >= w) continue;
    float out[3];
    EpfPixel<1>(f, s_in + (ly + 2) * P0 + (lx + 2), R0 * P0, P0, x, y, cs0, cs1, cs2, out);
#pragma unroll
    for (int c = 0; c < 3; c++) s_p1[(c * R1 + ly) * P1 + lx] = out[c];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < R1 * R1; i += 256) {
    const int ly = i / R1, lx = i - ly * R1;
    const int y = y0 + ly - 1, x = x0 + lx - 1;
    if (y >= 0 && y < h && x >= 0 && x < w) continue;
    const int my = MirrorD(y, h) - y0 + 1, mx = MirrorD(x, w) - x0 + 1;
    if (my < 0 || my >= R1 || mx < 0 || mx >= R1) continue;                     // (farther out than any pixel of the image reaches)
#pragma unroll
    for (int c = 0; c < 3; c++) s_p1[(c * R1 + ly) * P1 + lx] = s_p1[(c * R1 + my) * P1 + mx];
  }
  __syncthreads();
#pragma unroll 1
  for (int q = 0; q < 4; q++) {
    const int lx = threadIdx.x & 31, ly = (threadIdx.x >> 5) + 8 * q;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) continue;
    float out[3];
    EpfPixel<2>(f, s_p1 + (ly + 1) * P1 + (lx + 1), R1 * P1, P1, x, y, cs0, cs1, cs2, out);
    const float A = f.alpha_plane ? (float)f.alpha_plane[(size_t)y * f.width + x] * f.alpha_factor : 1.0f;
    ColorAndStore(f, x, y, out[0], out[1], out[2], A);
  }
}

// =====================================================================================================================
// Fused restoration + colour + write: gaborish -> EPF pass 1 -> XYB -> (s)RGB -> interleaved samples, one 64x32 pixel
// tile per 256-thread workgroup.  The XYB planes are read once (with a 3-pixel mirrored halo) into LDS, the gaborish
// result (tile + 2-pixel halo) lives in LDS, pixels go straight to the caller layout: 12 + C*bytes B/px of HBM traffic
// instead of 63.  Arithmetic and operation order are those of GaborishKernel / EpfKernel<1> / OutputKernel (mirroring
// the inputs of the symmetric 3x3 kernel gives exactly the gaborish value at the mirrored coordinate).
// =====================================================================================================================
#ifndef JXL_FTH
#define JXL_FTH 24
#endif
constexpr int kFtW = 32, kFtH = JXL_FTH;   // (JXL_FTH: tile height, an A/B knob — 16 gives 10 KB of LDS per workgroup instead of 14)
#ifndef JXL_FPAD
#define JXL_FPAD 1
#endif
constexpr int kFinW = kFtW + 6, kFinH = kFtH + 6, kFinP = kFinW + JXL_FPAD;      // input region incl. halo 3, padded pitch (JXL_FPAD 0: 13.4 instead of 13.7 KB per workgroup — a seventh one beside a 66 KB HF workgroup)
constexpr int kFgW = kFtW + 4, kFgH = kFtH + 4, kFgP = kFgW + 1;          // gaborish region incl. halo 2
constexpr size_t kFusedLds = (size_t)(3 * kFinH * kFinP) * sizeof(float);   // the gaborish tile reuses the input tile's LDS (14 KB)

// Workgroup -> tile: the dispatcher hands consecutive workgroups to the eight XCDs in turn, each with an L2 of its own, and a tile shares
// its halo (and the cache lines its 16-byte-aligned rows straddle) with its neighbours: with tile = workgroup id every shared line was
// fetched from HBM once per tile (rocprof: 280 MB read per 4K frame for 100 MB of planes).  Each XCD gets a contiguous run of the
// frame's tiles instead (bijective for any tile count), so neighbours meet in one L2.
__device__ __forceinline__ uint32_t XcdContiguous(uint32_t bid, uint32_t nwg) {
  const uint32_t q = nwg >> 3, r = nwg & 7u, c = bid & 7u;
  return (c < r ? c * (q + 1) : r * (q + 1) + (c - r) * q) + (bid >> 3);
}

__global__ __launch_bounds__(256) void FusedGabEpf1OutKernel(const FrameDev* __restrict__ frames, int unfused, int tiles_x, int swizzle) {
  const FrameDev& f = frames[blockIdx.z];
  if (f.is_modular || !FusedEligible(f, unfused)) return;
  const int w = (int)f.width, h = (int)f.height;
  const uint32_t tile = swizzle ? XcdContiguous(blockIdx.x, gridDim.x) : blockIdx.x;
  const int x0 = (int)(tile % (uint32_t)tiles_x) * kFtW, y0 = (int)(tile / (uint32_t)tiles_x) * kFtH;
  if (x0 >= w || y0 >= h) return;
  extern __shared__ __align__(16) float s_f[];
  float* s_in = s_f;                              // [3][kFinH][kFinP]
  float* s_gab = s_f;                             // [3][kFgH][kFgP] — overwrites the input tile (values pass through registers)
  const size_t stride = f.plane_stride;
  // ---- load input tile (+3 halo).  Tiles whose halo lies inside the image (all but the border ones) read aligned
  // 16-byte chunks — columns x0-4 .. x0+35, ten per row — and drop the two outer samples; the others mirror per sample.
  if (x0 >= 4 && y0 >= 3 && x0 + kFtW + 4 <= w && y0 + kFtH + 3 <= h) {
    constexpr int kV = (kFinW + 2) / 4;
    static_assert(kV * 4 == kFinW + 2 && kFtW % 4 == 0, "tile width");
    for (int i = threadIdx.x; i < kV * kFinH * 3; i += blockDim.x) {
      const int c = i / (kV * kFinH), r = i - c * (kV * kFinH), ly = r / kV, v = r - ly * kV;
      const float4 q = LdGKeep(reinterpret_cast<const float4*>(f.plane_a[c] + (size_t)(y0 + ly - 3) * stride + (x0 - 4 + v * 4)));
      float* d = s_in + (c * kFinH + ly) * kFinP + v * 4 - 1;      // q.x is the sample left of local column v * 4
      if (v > 0) d[0] = q.x;
      d[1] = q.y; d[2] = q.z;
      if (v < kV - 1) d[3] = q.w;
    }
  } else {
    for (int i = threadIdx.x; i < kFinW * kFinH; i += blockDim.x) {
      const int ly = i / kFinW, lx = i % kFinW;
      const int gx = MirrorD(x0 + lx - 3, w), gy = MirrorD(y0 + ly - 3, h);
      const size_t o = (size_t)gy * stride + gx;
#pragma unroll
      for (int c = 0; c < 3; c++) s_in[(c * kFinH + ly) * kFinP + lx] = LdG(f.plane_a[c] + o);
    }
  }
  __syncthreads();
  // ---- gaborish on tile + halo 2: every thread keeps its results in registers until all inputs have been read
  constexpr int kGabPerThread = (kFgW * kFgH + 255) / 256;
  float gv[kGabPerThread][3];
#pragma unroll
  for (int q = 0; q < kGabPerThread; q++) {
    const int i = (int)threadIdx.x + q * 256;
    if (i >= kFgW * kFgH) continue;
    const int ly = i / kFgW, lx = i % kFgW;
#pragma unroll
    for (int c = 0; c < 3; c++) {
      const float* t = s_in + (c * kFinH + ly) * kFinP + lx;        // row above (input coords are +1 relative to gab coords)
      const float* m = t + kFinP;
      const float* b = m + kFinP;
      const float sum0 = m[1];
      const float sum1 = (m[0] + m[2]) + (t[1] + b[1]);
      const float sum2 = (t[0] + t[2]) + (b[0] + b[2]);
      gv[q][c] = fmaf(sum2, f.gab_w[c * 3 + 2], fmaf(sum1, f.gab_w[c * 3 + 1], sum0 * f.gab_w[c * 3 + 0]));
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kGabPerThread; q++) {
    const int i = (int)threadIdx.x + q * 256;
    if (i >= kFgW * kFgH) continue;
    const int ly = i / kFgW, lx = i % kFgW;
#pragma unroll
    for (int c = 0; c < 3; c++) s_gab[(c * kFgH + ly) * kFgP + lx] = gv[q][c];
  }
  __syncthreads();
  // ---- EPF pass 1 + colour + store
#ifndef JXL_NO_PACKED_STORE
  // 1: u8 RGB, 2: u8 RGBA written as dwords (rows and buffer 4-byte aligned, width a multiple of 4 so that a quad of lanes is inside the image or outside it)
  const int packed = (f.out_type == 0 && f.out_orient <= 1 && (w & 3) == 0 && ((uintptr_t)f.out & 3) == 0 && (f.out_stride & 3) == 0 && f.img_w == f.width && f.img_h == f.height)
                         ? (f.out_channels == 3 ? 1 : f.out_channels == 4 ? 2 : 0) : 0;
#endif
  const float sm = f.epf_sm[1], bsm = f.epf_bsm[1];
  const float cs0 = f.epf_channel_scale[0], cs1 = f.epf_channel_scale[1], cs2 = f.epf_channel_scale[2];
  for (int i = threadIdx.x; i < kFtW * kFtH; i += blockDim.x) {
    const int ly = i / kFtW, lx = i % kFtW;
    const int x = x0 + lx, y = y0 + ly;
    if (x >= w || y >= h) continue;
    const float* g0 = s_gab + (0 * kFgH + ly + 2) * kFgP + lx + 2;
    const float* g1 = g0 + kFgH * kFgP;
    const float* g2 = g1 + kFgH * kFgP;
    float X = g0[0], Y = g1[0], B = g2[0];
    const float is = LdG(f.inv_sigma + (size_t)(y / 8) * f.bw + x / 8);
    if (!(is < -3.90524291751269967465540850526868f)) {
      const bool border = (x % 8 == 0) || (x % 8 == 7) || (y % 8 == 0) || (y % 8 == 7);
      const float vmul = is * (border ? bsm : sm);
      float wsum = 1.0f;
      float a0 = X, a1 = Y, a2 = B;
      // Sums of absolute differences over the plus-shaped neighbourhood for the four taps N, W, E, S (EpfKernel<1>'s
      // loops written out: same terms, same order).  |centre - neighbour| occurs in two taps each (tap d at offset 0,
      // tap -d at offset d) and is computed once: 16 differences per channel instead of 20.
      float sN[3], sW[3], sE[3], sS[3], vN[3], vW[3], vE[3], vS[3];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float* g = c == 0 ? g0 : c == 1 ? g1 : g2;
        const float P = g[0], N = g[-kFgP], W = g[-1], E = g[1], S = g[kFgP];
        const float NN = g[-2 * kFgP], NW = g[-kFgP - 1], NE = g[-kFgP + 1], WW = g[-2], EE = g[2], SW = g[kFgP - 1], SE = g[kFgP + 1], SS = g[2 * kFgP];
        const float dN = fabsf(N - P), dW = fabsf(W - P), dE = fabsf(E - P), dS = fabsf(S - P);
        // offsets in order (0,0), (0,-1), (-1,0), (1,0), (0,1)
        sN[c] = ((((0.f + dN) + fabsf(NN - N)) + fabsf(NW - W)) + fabsf(NE - E)) + dS;
        sW[c] = ((((0.f + dW) + fabsf(NW - N)) + fabsf(WW - W)) + dE) + fabsf(SW - S);
        sE[c] = ((((0.f + dE) + fabsf(NE - N)) + dW) + fabsf(EE - E)) + fabsf(SE - S);
        sS[c] = ((((0.f + dS) + dN) + fabsf(SW - W)) + fabsf(SE - E)) + fabsf(SS - S);
        vN[c] = N; vW[c] = W; vE[c] = E; vS[c] = S;
      }
#pragma unroll
      for (int t = 0; t < 4; t++) {
        const float* st = t == 0 ? sN : t == 1 ? sW : t == 2 ? sE : sS;
        const float* vt = t == 0 ? vN : t == 1 ? vW : t == 2 ? vE : vS;
        float sad = 0.f;
        sad = fmaf(st[0], cs0, sad);
        sad = fmaf(st[1], cs1, sad);
        sad = fmaf(st[2], cs2, sad);
        const float wgt = fmaxf(0.0f, fmaf(sad, vmul, 1.0f));
        wsum += wgt;
        a0 = fmaf(wgt, vt[0], a0);
        a1 = fmaf(wgt, vt[1], a1);
        a2 = fmaf(wgt, vt[2], a2);
      }
      const float inv = 1.0f / wsum;
      X = a0 * inv; Y = a1 * inv; B = a2 * inv;
    }
    // XYB -> linear -> sRGB (OutputKernel)
    const float gr = (Y + X) - f.neg_bias_cbrt[0];
    const float gg = (Y - X) - f.neg_bias_cbrt[1];
    const float gb = B - f.neg_bias_cbrt[2];
    const float mr = fmaf(gr * gr, gr, f.neg_bias[0]);
    const float mg = fmaf(gg * gg, gg, f.neg_bias[1]);
    const float mb = fmaf(gb * gb, gb, f.neg_bias[2]);
    float r = fmaf(f.opsin_inv[2], mb, fmaf(f.opsin_inv[1], mg, f.opsin_inv[0] * mr));
    float g = fmaf(f.opsin_inv[5], mb, fmaf(f.opsin_inv[4], mg, f.opsin_inv[3] * mr));
    float b = fmaf(f.opsin_inv[8], mb, fmaf(f.opsin_inv[7], mg, f.opsin_inv[6] * mr));
    if (f.color_mode == 0) { r = LinearToSrgb(r); g = LinearToSrgb(g); b = LinearToSrgb(b); }
    if (f.is_gray) r = g;
#ifndef JXL_NO_PACKED_STORE
    if (packed) {
      // u8 RGB / RGBA in image orientation: whole dwords instead of three or four byte stores per pixel.  RGB: the four lanes of a quad hold 12 bytes = three
      // dwords; lane j takes the rest of its own pixel and the head of its right neighbour's (DPP quad_perm [1, 2, 3, 3]) and lanes 0..2 store.  Same rounding
      // as StoreSample.
      const uint32_t pr = (uint32_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, r)) * f.out_int_mul) & 0xFFu;
      const uint32_t pg = (uint32_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, g)) * f.out_int_mul) & 0xFFu;
      const uint32_t pb = (uint32_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, b)) * f.out_int_mul) & 0xFFu;
      const uint32_t p = pr | (pg << 8) | (pb << 16);
      uint8_t* const row = f.out + (size_t)y * f.out_stride;
      if (packed == 2) {
        const float a = f.alpha_plane ? (float)f.alpha_plane[(size_t)y * f.width + x] * f.alpha_factor : 1.0f;
        const uint32_t pa = (uint32_t)__float2int_rn(fminf(1.0f, fmaxf(0.0f, a)) * f.out_int_mul) & 0xFFu;
        StoreOut32(reinterpret_cast<uint32_t*>(row + 4 * (size_t)x), p | (pa << 24));
      } else {
        const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)p, 0xF9, 0xF, 0xF, false);
        const uint32_t j = (uint32_t)lx & 3u;
        const uint32_t d = (p >> (8u * j)) | (nx << (24u - 8u * j));
        if (j != 3u) StoreOut32(reinterpret_cast<uint32_t*>(row + 3 * (size_t)x + j), d);
      }
      continue;
    }
#endif
    StorePixel(f, x, y, r, g, b, f.alpha_plane ? (float)f.alpha_plane[(size_t)y * f.width + x] * f.alpha_factor : 1.0f);
  }
}

// =====================================================================================================================
// K_mod*: Modular frames.  Global stream (channels that fit a group), per-group streams with local palette / RCT,
// global inverse transforms, integer -> sample conversion.
// =====================================================================================================================
__device__ void InvRctD(int32_t* c0, int32_t* c1, int32_t* c2, size_t n, uint32_t rct_type, size_t tid0, size_t tstride) {
  const uint32_t perm = rct_type / 7, kind = rct_type % 7;
  for (size_t i = tid0; i < n; i += tstride) {
    const int32_t a = c0[i], b = c1[i], c = c2[i];
    int32_t o0, o1, o2;
    if (kind == 6) {
      const int32_t tmp = (int32_t)((uint32_t)a - (uint32_t)(c >> 1));
      o1 = (int32_t)((uint32_t)c + (uint32_t)tmp);
      o2 = (int32_t)((uint32_t)tmp - (uint32_t)(b >> 1));
      o0 = (int32_t)((uint32_t)o2 + (uint32_t)b);
    } else {
      int32_t first = a, second = b, third = c;
      if (kind & 1) third = (int32_t)((uint32_t)third + (uint32_t)first);
      if ((kind >> 1) == 1) second = (int32_t)((uint32_t)second + (uint32_t)first);
      else if ((kind >> 1) == 2) second = (int32_t)((uint32_t)second + (uint32_t)(((int64_t)first + third) >> 1));
      o0 = first; o1 = second; o2 = third;
    }
    int32_t res[3];
    res[perm % 3] = o0; res[(perm + 1 + perm / 3) % 3] = o1; res[(perm + 2 - perm / 3) % 3] = o2;
    c0[i] = res[0]; c1[i] = res[1]; c2[i] = res[2];
  }
}

// palette.h kDeltaPalette: the implicit delta entries negative indices select (index -> entry (i + 1) / 2, sign by parity)
__constant__ int16_t d_delta_palette[72][3] = {
    {0, 0, 0},       {4, 4, 4},       {11, 0, 0},      {0, 0, -13},     {0, -12, 0},     {-10, -10, -10}, {-18, -18, -18}, {-27, -27, -27},
    {-18, -18, 0},   {0, 0, -32},     {-32, 0, 0},     {-37, -37, -37}, {0, -32, -32},   {24, 24, 45},    {50, 50, 50},    {-45, -24, -24},
    {-24, -45, -45}, {0, -24, -24},   {-34, -34, 0},   {-24, 0, -24},   {-45, -45, -24}, {64, 64, 64},    {-32, 0, -32},   {0, -32, 0},
    {-32, 0, 32},    {-24, -45, -24}, {45, 24, 45},    {24, -24, -45},  {-45, -24, 24},  {80, 80, 80},    {64, 0, 0},      {0, 0, -64},
    {0, -64, -64},   {-24, -24, 45},  {96, 96, 96},    {64, 64, 0},     {45, -24, -24},  {34, -34, 0},    {112, 112, 112}, {24, -45, -45},
    {45, 45, -24},   {0, -32, 32},    {24, -24, 45},   {0, 96, 96},     {45, -24, 24},   {24, -45, -24},  {-24, -45, 24},  {0, -64, 0},
    {96, 0, 0},      {128, 128, 128}, {64, 0, 64},     {144, 144, 144}, {96, 96, 0},     {-36, -36, 36},  {45, -24, -45},  {45, -45, -24},
    {0, 0, -96},     {0, 128, 128},   {0, 96, 0},      {45, 24, -45},   {-128, 0, 0},    {24, -45, 24},   {-45, 24, -45},  {64, 0, -64},
    {64, -64, -64},  {96, 0, 96},     {45, -45, 24},   {24, 45, -45},   {64, 64, -64},   {128, 128, 0},   {0, 0, -128},    {-24, 45, -45}};

__device__ __forceinline__ int32_t PaletteValue(const int32_t* pal, int pal_w, int index, int c, int bit_depth) {
  // palette.h GetPaletteValue
  if (index < 0) {
    if (c >= 3) return 0;
    index = -(index + 1);
    index %= 1 + 2 * (72 - 1);
    int32_t r = (int32_t)d_delta_palette[(index + 1) >> 1][c] * ((index & 1) ? 1 : -1);
    if (bit_depth > 8) r *= 1 << (bit_depth - 8);
    return r;
  }
  if (index < pal_w) return pal[(size_t)c * pal_w + index];
  if (c >= 3) return 0;
  if (index < pal_w + 64) {
    int i = (index - pal_w) >> (c * 2);
    return (int32_t)(((int64_t)(i % 4) * ((1 << bit_depth) - 1)) / 4 + (1 << max(0, bit_depth - 3)));
  }
  int i = index - pal_w - 64;
  for (int k = 0; k < c; k++) i /= 5;
  return (int32_t)(((int64_t)(i % 5) * ((1 << bit_depth) - 1)) / 4);
}

// The global stream (meta channels + every channel that fits one group) with the same cooperative decoder: one wavefront
// per frame.  Replaces the one-thread ModularGlobalKernel whenever the stream uses the global tree.
__global__ __launch_bounds__(64) void ModularGlobalFastKernel(const FrameDev* __restrict__ frames, uint32_t tree_cap, uint32_t lds_bytes, uint32_t wp_base) {
  const FrameDev& f = frames[blockIdx.x];
  if (f.mod_nchan == 0) return;
  ModTables T;
  // the global stream may carry a tree and a code of its own (GroupHeader.use_global_tree = 0; f.mod_global_bitpos is then past them)
  const ModLocalDev* local = f.mod_local && f.mod_local[0].tree ? &f.mod_local[0] : nullptr;
  const TreeNode* const tree = local ? local->tree : f.tree;
  const uint32_t tree_nodes = local ? local->tree_nodes : f.tree_nodes;
  const DevCode& code = local ? local->code : f.mod_code;
  StageModular(tree, tree_nodes, code, T, tree_cap, lds_bytes);
  if (wp_base) T.wp_off = wp_base;
  const uint32_t lane = threadIdx.x & 63;
  BitReaderP br;
  br.Init(f.cs, f.mod_global_bitpos, f.cs_size);
  uint32_t state = 0x130000u;
  if (lane == 0 && !code.use_prefix) state = br.Read(32);
  ModularCtx mc;
  mc.tree = tree; mc.code = &code; mc.uses_wp = local ? local->uses_wp : f.uses_wp; mc.wp = f.gwp; mc.stream_id = 0; mc.narrow_wp = f.mod_bits <= 12;
  mc.max_prop = local ? local->max_prop : f.tree_max_prop;
  mc.wp_scratch = f.mod_wp_scratch; mc.wp_scratch_ints = f.mod_wp_stride; mc.status = f.status;
  mc.slow = code.use_prefix || code.lz77;
  __shared__ ModRefs s_refs;
  mc.refs = &s_refs;
  Lz77State lz;
  if (code.lz77) {
    uint32_t dist_mult = 0;
    for (uint32_t c = 0; c < f.mod_global_decodable; c++) dist_mult = max(dist_mult, f.mod_chan[c].w);
    lz.Init(f.lz_window, dist_mult);
    mc.lz = &lz;
  }
  for (uint32_t c = 0; c < f.mod_global_decodable; c++) {
    const ModChanDev mcd = f.mod_chan[c];
    if (mcd.w == 0 || mcd.h == 0) continue;  // (empty channels keep their index: property 0 is the position in the list)
    ChannelDesc ch;
    ch.data = ModPlane(f, mcd); ch.w = (int)mcd.w; ch.h = (int)mcd.h; ch.stride = (int)mcd.w;
    if (mc.max_prop >= 16) {   // reference channels: earlier channels of this stream with the same geometry, nearest first
      if (lane == 0) {
        int n = 0;
        for (int j = (int)c - 1; j >= 0 && n < kMaxModRefs; j--) {
          const ModChanDev r = f.mod_chan[j];
          if (r.w != mcd.w || r.h != mcd.h || r.hshift != mcd.hshift || r.vshift != mcd.vshift) continue;
          s_refs.data[n] = ModPlane(f, r); s_refs.stride[n] = (int)r.w; n++;
        }
        s_refs.n = n;
      }
      WaveSync();
    }
    DecodeChannelCoop<2>(br, state, T, mc, ch, (int)c);
  }
  if (lane == 0) {
    if (state != 0x130000u) SetError(f, kErrAnsFinalState);
    else if (br.BitPos() > f.cs_size * 8) SetError(f, kErrOverrun);
    else f.stream_end_bitpos[1] = br.BitPos();
  }
}

// Per-section Modular sub-streams (dec_modular.cc DecodeGroup) through the cooperative wavefront decoder of the LF stage:
// unit < num_lf_groups → the ModularLfGroup stream of that LF group (channels squeezed by >= 3 in both directions, stream
// id 1+nlf+g); otherwise the pass-group stream (shift 0..2, stream id 1+3nlf+17+g).  One wavefront per unit, four per
// workgroup sharing the LDS copy of the MA tree and the entropy code.  Streams without local transforms (squeezed / plain
// channels, alpha of VarDCT frames) are decoded straight into the frame planes; streams with local palettes / RCTs
// (bench.jxl) go through a per-unit scratch, the wavefront undoes the transforms and copies the rectangles out.
constexpr int kMaxXformChan = 8;
struct ModUnitShared {
  int go, nch, direct;
  unsigned long long used;
  GroupHeaderD gh;
  ChannelDesc ch[12], dst[kMaxXformChan], fresh[4];
};
// local_pass = 1: the launch for units whose stream carries a tree / code of its own (f.mod_local) — one wavefront per workgroup,
// each staging its unit's tables; the regular launch (0) skips those units.
#ifndef JXL_MODGROUP_MINW
#define JXL_MODGROUP_MINW 2     // wavefronts per SIMD the register budget of ModularGroupFastKernel allows (one serial chain per wavefront: residency is throughput)
#endif
// BIGTREE: the instantiation for frames whose MA tree has more than kBigTreeNodes nodes (subtrees of hundreds of splits per stream: the wave-wide decoder's big-tree form, at one
// wavefront per SIMD — such trees are copied per wavefront, two wavefronts per workgroup, and the LDS admits no more anyway)
template <bool BIGTREE> __global__ __launch_bounds__(256, BIGTREE ? 1 : JXL_MODGROUP_MINW) void ModularGroupFastKernel(const FrameDev* __restrict__ frames, uint32_t tree_cap, uint32_t lds_bytes, uint32_t wp_base, uint32_t local_pass) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.mod_nchan == 0 || f.single_section) return;
  if (local_pass ? !f.mod_local : !f.tree) return;   // (a frame without a global tree: every unit is decoded by the local pass)
  const uint32_t total = f.num_lf_groups + f.num_groups * f.mod_unit_passes;
  const uint32_t nwaves = blockDim.x >> 6;          // 4, or 2 when every wavefront needs a large pruned-tree slice; 1 in the local pass
  if (blockIdx.x * nwaves >= total) return;
  const uint32_t first = f.mod_global_decodable;
  if (first >= f.mod_nchan) return;
  const ModLocalDev* local = nullptr;
  if (local_pass) { local = &f.mod_local[1 + blockIdx.x]; if (!local->tree) return; }
  const TreeNode* const tree = local ? local->tree : f.tree;
  const uint32_t tree_nodes = local ? local->tree_nodes : f.tree_nodes;
  const DevCode& code = local ? local->code : f.mod_code;
  ModTables T;
  StageModular(tree, tree_nodes, code, T, tree_cap, lds_bytes);
  const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (wp_base) T.wp_off = wp_base + wave * kWpLdsBytes;
  const uint32_t unit = blockIdx.x * nwaves + wave;
  if (unit >= total) return;                     // (no block-wide barrier after this point)
  if (!local_pass && f.mod_local && f.mod_local[1 + unit].tree) return;   // decoded by the local pass
  const bool is_lf = unit < f.num_lf_groups;
  if (!f.is_modular && is_lf) return;            // VarDCT: extra channels are never squeezed here, so ModularLfGroup is empty
  // units after the LF groups: PassGroup (pass, g), pass-major — Modular frames: every pass; VarDCT frames: the pass that carries the extra channels
  const uint32_t last_pass = is_lf ? 0 : f.is_modular ? (unit - f.num_lf_groups) / f.num_groups : f.mod_pass;
  const uint32_t g = is_lf ? unit : (unit - f.num_lf_groups) % f.num_groups;
  const uint32_t dim = is_lf ? f.group_dim * 8 : f.group_dim;
  const uint32_t cols = is_lf ? f.xlfgroups : f.xgroups;
  const uint32_t x0 = (g % cols) * dim, y0 = (g / cols) * dim;
  const int min_shift = is_lf ? 3 : f.pass_min_shift[last_pass], max_shift = is_lf ? 1000 : f.pass_max_shift[last_pass];
  __shared__ ModUnitShared s_unit[kLfWaves];
  ModUnitShared& U = s_unit[wave];
  const uint32_t si = is_lf ? 1 + g : 2 + f.num_lf_groups + last_pass * f.num_groups + g;
  const uint64_t sec_end = f.sec_off[si] + f.sec_size[si];
  BitReaderP br;
  br.Init(f.cs, f.sec_off[si] * 8, sec_end);
  uint32_t state = 0;
  ChannelDesc d;
  int32_t* scratch = f.mod_group_scratch + (uint64_t)unit * f.mod_group_scratch_stride;
  if (lane == 0) {
    int nch = 0;
    for (uint32_t c = first; c < f.mod_nchan; c++) nch += ModUnitRect(f, c, x0, y0, dim, min_shift, max_shift, &d) ? 1 : 0;
    U.go = 0; U.nch = 0; U.direct = 1; U.used = 0;
    if (nch > 0) {
      BitReader tmp;
      tmp.Init(f.cs, f.is_modular ? f.sec_off[si] * 8 : f.hf_end_bitpos[g], f.cs_size);
      bool ok = ReadGroupHeader(tmp, U.gh) && (U.gh.use_global_tree ? local == nullptr : local != nullptr);   // (a local tree the host has not parsed: VarDCT frames)
      if (ok && U.gh.ntransforms != 0) {
        // local transforms: channels are decoded into the unit's scratch (at most kMaxXformChan of them)
        U.direct = 0;
        if (nch > kMaxXformChan) ok = false;
        unsigned long long used = 0;
        int k = 0;
        for (uint32_t c = first; ok && c < f.mod_nchan; c++) if (ModUnitRect(f, c, x0, y0, dim, min_shift, max_shift, &d)) {
          U.dst[k] = d;
          U.ch[k].data = scratch + used; U.ch[k].w = d.w; U.ch[k].h = d.h; U.ch[k].stride = d.w; U.ch[k].hs = d.hs; U.ch[k].vs = d.vs;
          used += (unsigned long long)d.w * d.h;
          k++;
        }
        int nmeta = 0;
        for (uint32_t i = 0; ok && i < U.gh.ntransforms; i++) {           // MetaApply on the channel list
          auto& t = U.gh.t[i];
          if (t.id == 0) { if (t.begin_c + 3 > (uint32_t)nch) ok = false; }
          else if (t.id == 1) {
            const uint32_t endc = t.begin_c + t.num_c - 1;
            if (endc >= (uint32_t)nch || (int)t.begin_c < nmeta || nch + 1 - (int)(t.num_c - 1) > 12 || t.nb_colors > 65536 || t.num_c > 4) { ok = false; break; }
            for (uint32_t q = endc + 1; q < (uint32_t)nch; q++) U.ch[q - (t.num_c - 1)] = U.ch[q];     // drop channels begin_c+1..endc
            nch -= (int)(t.num_c - 1);
            for (int q = nch; q > 0; q--) U.ch[q] = U.ch[q - 1];                                         // palette channel at index 0
            nch++;
            U.ch[0].data = scratch + used; U.ch[0].w = (int)t.nb_colors; U.ch[0].h = (int)t.num_c; U.ch[0].stride = (int)t.nb_colors; U.ch[0].hs = -1; U.ch[0].vs = 0;
            used += (unsigned long long)t.nb_colors * t.num_c;
            nmeta++;
          } else ok = false;                                               // local squeeze
        }
        if (used > f.mod_group_scratch_stride) ok = false;
        U.used = used;
      }
      if (!ok) SetError(f, kErrUnsupported);
      else {
        br.Init(f.cs, local ? local->data_bitpos : tmp.BitPos(), sec_end);   // (local: past the stream's own tree and code)
        state = code.use_prefix ? 0x130000u : br.Read(32);
        U.nch = nch;
        U.go = 1;
      }
    }
  }
  WaveSync();
  if (!U.go) return;
  ModularCtx mc;
  mc.tree = tree; mc.code = &code; mc.uses_wp = local ? local->uses_wp : f.uses_wp; mc.wp = U.gh.wp; mc.narrow_wp = f.mod_bits <= 12;
  mc.max_prop = local ? local->max_prop : f.tree_max_prop;
  mc.stream_id = is_lf ? 1 + f.num_lf_groups + g : 1 + 3 * f.num_lf_groups + 17 + last_pass * f.num_groups + g;
  mc.wp_scratch = f.mod_wp_scratch + (uint64_t)(1 + unit) * f.mod_wp_stride; mc.wp_scratch_ints = f.mod_wp_stride; mc.status = f.status;
  mc.slow = code.use_prefix || code.lz77;
  __shared__ ModRefs s_refs_w[kLfWaves];
  ModRefs& refs = s_refs_w[wave];
  mc.refs = &refs;
  Lz77State lz;
  if (code.lz77) {
    uint32_t dist_mult = 0;     // widest channel of the stream (modular/encoding/encoding.cc)
    if (U.direct) { for (uint32_t c = first; c < f.mod_nchan; c++) if (ModUnitRect(f, c, x0, y0, dim, min_shift, max_shift, &d)) dist_mult = max(dist_mult, (uint32_t)d.w); }
    else for (int c = 0; c < U.nch; c++) dist_mult = max(dist_mult, (uint32_t)U.ch[c].w);
    lz.Init(f.lz_window + (uint64_t)(1 + unit) * Lz77State::kWindow, dist_mult);
    mc.lz = &lz;
  }
  if (U.direct) {
    int k = 0;
    for (uint32_t c = first; c < f.mod_nchan; c++) if (ModUnitRect(f, c, x0, y0, dim, min_shift, max_shift, &d)) {
      if (mc.max_prop >= 16) {   // reference channels: earlier channels of this stream with the same geometry, nearest first
        if (lane == 0) {
          int n = 0;
          ChannelDesc r;
          for (int j = (int)c - 1; j >= (int)first && n < kMaxModRefs; j--) {
            if (!ModUnitRect(f, (uint32_t)j, x0, y0, dim, min_shift, max_shift, &r)) continue;
            if (r.w != d.w || r.h != d.h || r.hs != d.hs || r.vs != d.vs) continue;
            refs.data[n] = r.data; refs.stride[n] = r.stride; n++;
          }
          refs.n = n;
        }
        WaveSync();
      }
      DecodeChannelCoop<BIGTREE ? 2 : 1>(br, state, T, mc, d, k++);
    }
  } else {
    const int nch = U.nch;
    for (int c = 0; c < nch; c++) {
      const ChannelDesc cd = U.ch[c];
      if (mc.max_prop >= 16) {
        if (lane == 0) {
          int n = 0;
          for (int j = c - 1; j >= 0 && n < kMaxModRefs; j--) {
            const ChannelDesc r = U.ch[j];
            if (r.w != cd.w || r.h != cd.h || r.hs != cd.hs || r.vs != cd.vs) continue;
            refs.data[n] = r.data; refs.stride[n] = r.stride; n++;
          }
          refs.n = n;
        }
        WaveSync();
      }
      DecodeChannelCoop<BIGTREE ? 2 : 1>(br, state, T, mc, cd, c);
    }
  }
  int fail = 0;
  if (lane == 0) {
    if (state != 0x130000u) { SetError(f, kErrAnsFinalState); fail = 1; }
    else if (br.BitPos() > sec_end * 8) { SetError(f, kErrOverrun); fail = 1; }
    U.go = !fail;
  }
  WaveSync();
  if (U.direct || !U.go) return;
  // ---- undo the local transforms (reverse order), the whole wavefront
  int nch = U.nch;
  for (int i = (int)U.gh.ntransforms - 1; i >= 0; i--) {
    const auto t = U.gh.t[i];
    if (t.id == 0) {
      const ChannelDesc a = U.ch[t.begin_c], b = U.ch[t.begin_c + 1], c = U.ch[t.begin_c + 2];
      InvRctD(a.data, b.data, c.data, (size_t)a.w * a.h, t.rct_type, lane, 64);
      WaveSync();
    } else {
      // inverse palette: channel 0 = palette, channel begin_c+1 = indices -> num_c channels (fresh storage at the scratch tail)
      const ChannelDesc pal = U.ch[0];
      const ChannelDesc idx = U.ch[t.begin_c + 1];
      const size_t n = (size_t)idx.w * idx.h;
      if (lane == 0) {
        if (U.used + (unsigned long long)(t.num_c - 1) * n > f.mod_group_scratch_stride) { SetError(f, kErrUnsupported); U.go = 0; }
        else {
          int32_t* tail = scratch + U.used;
          for (uint32_t c = 1; c < t.num_c; c++) { U.fresh[c] = idx; U.fresh[c].data = tail + (size_t)(c - 1) * n; }
          U.fresh[0] = idx;
          U.used += (unsigned long long)(t.num_c - 1) * n;
        }
      }
      WaveSync();
      if (!U.go) return;
      const int bit_depth = min((int)f.mod_bits, 24);
      for (size_t k = lane; k < n; k += 64) {
        const int index = idx.data[k];
        for (int c = (int)t.num_c - 1; c >= 0; c--) U.fresh[c].data[k] = PaletteValue(pal.data, pal.w, index, c, bit_depth);
      }
      WaveSync();
      if (lane == 0) {   // channel list: drop the palette (0), replace the index channel by num_c channels
        ChannelDesc tmp[12];
        int m = 0;
        for (int k = 1; k < nch; k++) {
          if ((uint32_t)(k - 1) == t.begin_c) { for (uint32_t c = 0; c < t.num_c; c++) tmp[m++] = U.fresh[c]; }
          else tmp[m++] = U.ch[k];
        }
        for (int k = 0; k < m; k++) U.ch[k] = tmp[k];
        U.nch = m;
      }
      WaveSync();
      nch = U.nch;
    }
  }
  // ---- copy into the frame planes
  for (int k = 0; k < nch && k < kMaxXformChan; k++) {
    const ChannelDesc cd = U.ch[k], dst = U.dst[k];
    for (size_t i = lane; i < (size_t)cd.w * cd.h; i += 64) {
      const size_t yy = i / cd.w, xx = i % cd.w;
      dst.data[yy * dst.stride + xx] = cd.data[i];
    }
  }
}

// ---- inverse Squeeze (squeeze.cc InvHSqueeze / InvVSqueeze; ISO/IEC 18181-1 "smooth tendency") ----------------------
__device__ __forceinline__ int64_t SqueezeTendency(int64_t B, int64_t a, int64_t n) {
  int64_t diff = 0;
  if (B >= a && a >= n) {
    diff = (4 * B - 3 * n - a + 6) / 12;
    if (diff - (diff & 1) > 2 * (B - a)) diff = 2 * (B - a) + 1;
    if (diff + (diff & 1) > 2 * (a - n)) diff = 2 * (a - n);
  } else if (B <= a && a <= n) {
    diff = (4 * B - 3 * n - a - 6) / 12;
    if (diff + (diff & 1) < 2 * (B - a)) diff = 2 * (B - a) - 1;
    if (diff - (diff & 1) < 2 * (a - n)) diff = 2 * (a - n);
  }
  return diff;
}
__device__ __forceinline__ void SqueezePair(int64_t prev, int64_t a, int64_t next, int64_t dmt, int32_t* o0, int32_t* o1) {
  const int64_t diff = dmt + SqueezeTendency(prev, a, next);
  const int64_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
  *o0 = (int32_t)A; *o1 = (int32_t)(A - diff);
}
// The same pair in 32-bit arithmetic when the four inputs are small enough for 4 B - 3 n - a +- 6 and 2 a + diff to stay inside an int32 (every image of up to 24 bits per
// sample; the 64-bit form — an emulated 64-bit division per pair — stays for streams that carry larger values)
__device__ __forceinline__ void SqueezePairFast(int32_t prev, int32_t a, int32_t next, int32_t dmt, int32_t* o0, int32_t* o1) {
  auto mag = [](int32_t v) { return (uint32_t)(v ^ (v >> 31)); };       // |v| (or |v| - 1): a bound, no overflow for INT_MIN
  if (__builtin_expect((mag(prev) | mag(a) | mag(next) | mag(dmt)) >= (1u << 26), 0)) { SqueezePair(prev, a, next, dmt, o0, o1); return; }
  const int32_t B = prev, n = next;
  int32_t t = 0;
  if (B >= a && a >= n) {
    t = (4 * B - 3 * n - a + 6) / 12;
    if (t - (t & 1) > 2 * (B - a)) t = 2 * (B - a) + 1;
    if (t + (t & 1) > 2 * (a - n)) t = 2 * (a - n);
  } else if (B <= a && a <= n) {
    t = (4 * B - 3 * n - a - 6) / 12;
    if (t + (t & 1) < 2 * (B - a)) t = 2 * (B - a) - 1;
    if (t - (t & 1) < 2 * (a - n)) t = 2 * (a - n);
  }
  const int32_t diff = dmt + t;
  const int32_t A = ((a * 2) + diff + (diff > 0 ? -(diff & 1) : (diff & 1))) >> 1;
  *o0 = A; *o1 = A - diff;
}
// horizontal: the recurrence runs along x, so one thread owns one row.  The averages and residuals of the next eight pairs do not depend on the recurrence: they are loaded
// ahead of it (round 5: one dependent memory round trip per pair before — 0.84 us per pair, 20 ms of inverse Squeeze per two 8192x8192 frames)
constexpr int kSqueezeAhead = 8;
// (blockIdx.y = the image: the same step of up to kSqueezeBatch equally shaped images in one launch — a step is one thread per row / column, 32 workgroups for the last one of an 8192 x 8192 channel)
__global__ __launch_bounds__(64) void ModInvSqueezeHKernel(const SqueezeBatch b, uint32_t aw, uint32_t rw, uint32_t h) {
  const uint32_t y = blockIdx.x * blockDim.x + threadIdx.x;
  if (y >= h) return;
  const int32_t* __restrict__ avg = b.avg[blockIdx.y]; const int32_t* __restrict__ res = b.res[blockIdx.y]; int32_t* __restrict__ out = b.out[blockIdx.y];
  const int32_t* pa = avg + (size_t)y * aw;
  const int32_t* pr = res + (size_t)y * rw;
  int32_t* po = out + (size_t)y * (aw + rw);
  int32_t a = aw ? pa[0] : 0, left = a;
  uint32_t x = 0;
  for (; x + kSqueezeAhead < rw; x += kSqueezeAhead) {        // (x + k + 1 <= rw - 1 < aw for every k)
    int32_t nx[kSqueezeAhead], rr[kSqueezeAhead];
#pragma unroll
    for (int k = 0; k < kSqueezeAhead; k++) { nx[k] = LdG(pa + x + k + 1); rr[k] = LdG(pr + x + k); }
#pragma unroll
    for (int k = 0; k < kSqueezeAhead; k++) {
      int32_t o0, o1;
      SqueezePairFast(left, a, nx[k], rr[k], &o0, &o1);
      po[2 * (x + k)] = o0; po[2 * (x + k) + 1] = o1;
      left = o1; a = nx[k];
    }
  }
  for (; x < rw; x++) {
    const int32_t next = x + 1 < aw ? pa[x + 1] : a;
    int32_t o0, o1;
    SqueezePairFast(left, a, next, pr[x], &o0, &o1);
    po[2 * x] = o0; po[2 * x + 1] = o1;
    left = o1; a = next;
  }
  if (aw > rw) po[2 * rw] = pa[rw];
}
// vertical: one thread per column, rows top to bottom (coalesced across the wave), eight rows loaded ahead of the recurrence
__global__ __launch_bounds__(256) void ModInvSqueezeVKernel(const SqueezeBatch b, uint32_t w, uint32_t ah, uint32_t rh) {
  const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= w) return;
  const int32_t* __restrict__ avg = b.avg[blockIdx.y]; const int32_t* __restrict__ res = b.res[blockIdx.y]; int32_t* __restrict__ out = b.out[blockIdx.y];
  int32_t a = ah ? avg[x] : 0, top = a;
  uint32_t y = 0;
  for (; y + kSqueezeAhead < rh; y += kSqueezeAhead) {        // (y + k + 1 <= rh - 1 < ah for every k)
    int32_t nx[kSqueezeAhead], rr[kSqueezeAhead];
#pragma unroll
    for (int k = 0; k < kSqueezeAhead; k++) { nx[k] = LdG(avg + (size_t)(y + k + 1) * w + x); rr[k] = LdG(res + (size_t)(y + k) * w + x); }
#pragma unroll
    for (int k = 0; k < kSqueezeAhead; k++) {
      int32_t o0, o1;
      SqueezePairFast(top, a, nx[k], rr[k], &o0, &o1);
      out[(size_t)(2 * (y + k)) * w + x] = o0; out[(size_t)(2 * (y + k) + 1) * w + x] = o1;
      top = o1; a = nx[k];
    }
  }
  for (; y < rh; y++) {
    const int32_t next = y + 1 < ah ? avg[(size_t)(y + 1) * w + x] : a;
    int32_t o0, o1;
    SqueezePairFast(top, a, next, res[(size_t)y * w + x], &o0, &o1);
    out[(size_t)(2 * y) * w + x] = o0; out[(size_t)(2 * y + 1) * w + x] = o1;
    top = o1; a = next;
  }
  if (ah > rh) out[(size_t)(2 * rh) * w + x] = avg[(size_t)rh * w + x];
}

// ---- global inverse transforms and the integer write stage (explicit arguments; launched per frame by the host) ----
__global__ void ModRctKernel(int32_t* a, int32_t* b, int32_t* c, size_t n, uint32_t rct_type) {
  InvRctD(a, b, c, n, rct_type, (size_t)blockIdx.x * blockDim.x + threadIdx.x, (size_t)gridDim.x * blockDim.x);
}

struct ModPaletteArgs { const int32_t* pal; int32_t* out[4]; uint32_t nb_colors, num_c, bit_depth; size_t n; };
__global__ void ModPaletteKernel(ModPaletteArgs a) {
  const size_t ts = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.n; i += ts) {
    const int index = a.out[0][i];
    for (int c = (int)a.num_c - 1; c >= 0; c--) a.out[c][i] = PaletteValue(a.pal, (int)a.nb_colors, index, c, (int)a.bit_depth);
  }
}

// Palettes with delta entries / a predictor (palette.h InvPalette, the nb_deltas / predictor branch): entries below nb_deltas and
// the implicit negative ones are added to a prediction from the already reconstructed neighbours of the same output channel, so
// a channel is a serial raster scan.  Lane c of one wavefront owns channel c; the lanes walk the pixels together because
// channel 0 is written over the index plane.
struct ModPaletteDeltaArgs { const int32_t* pal; int32_t* out[4]; uint32_t nb_colors, num_c, bit_depth, nb_deltas, predictor, w, h; WPHeader wp; int32_t* wp_scratch; uint32_t wp_stride; };
__global__ __launch_bounds__(64) void ModPaletteDeltaKernel(ModPaletteDeltaArgs a) {
  const uint32_t c = threadIdx.x;
  const bool active = c < a.num_c;
  int32_t* p = a.out[active ? c : 0];
  WPState wps;
  if (active && a.predictor == 6) wps.Init(a.wp_scratch + (size_t)c * a.wp_stride, (int32_t)a.w);
  const int w = (int)a.w, h = (int)a.h;
  for (int y = 0; y < h; y++) {
    for (int x = 0; x < w; x++) {
      const size_t i = (size_t)y * w + x;
      const int index = a.out[0][i];
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();           // every lane has the index before lane 0 replaces it
      if (active) {
        int32_t val = PaletteValue(a.pal, (int)a.nb_colors, index, (int)c, (int)a.bit_depth);
        const int64_t W = x ? p[i - 1] : (y ? p[i - w] : 0);
        const int64_t N = y ? p[i - w] : W;
        const int64_t NW = (x && y) ? p[i - w - 1] : W;
        const int64_t NE = (x + 1 < w && y) ? p[i - w + 1] : N;
        const int64_t WW = x > 1 ? p[i - 2] : W;
        const int64_t NN = y > 1 ? p[i - 2 * (size_t)w] : N;
        const int64_t NEE = (x + 2 < w && y) ? p[i - w + 2] : NE;
        int64_t wp_pred = 0;
        int32_t unused;
        if (a.predictor == 6) wp_pred = wps.Predict(a.wp, x, y, N, W, NE, NW, NN, &unused);
        if (index < (int)a.nb_deltas) {
          int64_t guess;
          switch (a.predictor) {
            case 0: guess = 0; break;
            case 1: guess = W; break;
            case 2: guess = N; break;
            case 3: guess = (W + N) / 2; break;
            case 4: { const int64_t pp = W + N - NW; guess = Abs64(pp - W) < Abs64(pp - N) ? W : N; break; }
            case 5: { const int64_t m = Min64(N, W), M = Max64(N, W); guess = NW < m ? M : (NW > M ? m : N + W - NW); break; }
            case 6: guess = (wp_pred + 3) >> 3; break;
            case 7: guess = NE; break;
            case 8: guess = NW; break;
            case 9: guess = WW; break;
            case 10: guess = (W + NW) / 2; break;
            case 11: guess = (N + NW) / 2; break;
            case 12: guess = (N + NE) / 2; break;
            default: guess = (6 * N - 2 * NN + 7 * W + WW + NEE + 3 * NE + 8) / 16; break;
          }
          val = (int32_t)((int64_t)val + guess);
        }
        p[i] = val;
        if (a.predictor == 6) wps.Update(val, x, y);
      }
    }
  }
}

__global__ void ModularOutputKernel(const FrameDev* __restrict__ frames, int fidx, ModOutputArgs a) {
  const FrameDev& f = frames[fidx];
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y * blockDim.y + threadIdx.y;
  if (x >= (int)f.width || y >= (int)f.height) return;
  const size_t o = (size_t)y * f.width + x;
  float r, g, b;
  auto sample = [&](int c) { const int32_t v = a.color[c][o]; return a.float_bits ? IntToFloatSample(v, a.float_bits, a.float_exp_bits) : (float)v * a.color_factor; };
  if (a.ncolor == 1) { r = g = b = sample(0); }
  else { r = sample(0); g = sample(1); b = sample(2); }
  const float al = a.alpha ? (float)a.alpha[o] * a.alpha_factor : 1.0f;
  // StorePixel picks r for gray output of gray images and g otherwise
  const uint32_t bps = f.out_type == 0 ? 1 : f.out_type == 2 ? 4 : 2;
  uint8_t* p = OutPixelPtr(f, x, y, bps);
  const uint32_t nc = f.out_channels;
  if (nc <= 2) { StoreSample(f, p, a.ncolor == 1 ? r : g); if (nc == 2) StoreSample(f, p + bps, al); }
  else { StoreSample(f, p, r); StoreSample(f, p + bps, g); StoreSample(f, p + 2 * bps, b); if (nc == 4) StoreSample(f, p + 3 * bps, al); }
}

// =====================================================================================================================
// launchers
// =====================================================================================================================
const char* const kKernelNames[] = {"LfDecodeKernel", "LfDequantKernel", "LfSmoothKernel", "LlfSigmaKernel", "HfDecodeKernel", "IdctKernel",
                                    "GaborishKernel", "EpfTileKernel", "OutputKernel", "ModularGlobalFastKernel", "ModularGroupFastKernel", nullptr};

// (per device, once; decoders of several host threads may get here at the same time)
static std::mutex g_tables_mu;
static bool g_tables_ready[64] = {false};
void InitDeviceTables(void* stream) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(g_tables_mu);
  if (g_tables_ready[dev]) return;
  static float wc[9][128];
  static float rs[6][32];
  for (int l = 1; l <= 8; l++) { const int N = 1 << l; for (int i = 0; i < N / 2; i++) wc[l][i] = (float)(1.0 / (2.0 * cos((i + 0.5) * M_PI / N))); }
  for (int l = 0; l < 6; l++) { const int N = 1 << l; for (int k = 0; k < N; k++) rs[l][k] = k == 0 ? 1.0f : (float)(sin(k * M_PI / (2.0 * N)) / sin(k * M_PI / (16.0 * N)) / 8.0); }
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_wc), wc, sizeof(wc), 0, hipMemcpyHostToDevice, (hipStream_t)stream);
  (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(d_resample), rs, sizeof(rs), 0, hipMemcpyHostToDevice, (hipStream_t)stream);
  (void)hipStreamSynchronize((hipStream_t)stream);
  g_tables_ready[dev] = true;
}

static inline int DivUp(int a, int b) { return (a + b - 1) / b; }

// Pipelined callers make the LF stage of a later batch runnable at the same moment as the HF stage of the current one (both
// wait for the previous step's end).  If the LF workgroups are dispatched first they pile up three per CU, the HF
// workgroups (80 KB of LDS) do not fit there until an LF workgroup ends, and that HF stage takes 82 instead of 44 ms (seen
// in one step out of three in a kernel trace).  So a large LF launch is preceded by this one-wavefront kernel, which
// holds the stream until an HF launch enqueued after it... (numbered `epoch` or later) has all its workgroups resident —
// with an HF workgroup on every CU only one more LF workgroup fits, and the dispatcher has to spread the launch — or,
// when no such launch shows up (different calling pattern), for 2 ms.
__global__ void HeadStartKernel(const uint32_t* __restrict__ sync, uint32_t epoch, uint64_t timeout_ticks) {
  const uint64_t t0 = wall_clock64();   // 100 MHz
  while ((int32_t)(__hip_atomic_load(sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - epoch) < 0 && wall_clock64() - t0 < timeout_ticks) __builtin_amdgcn_s_sleep(32);
}
// {workgroups started, last fully resident HF launch} per device + the number of HF launches enqueued so far
static std::mutex g_hf_sync_mu;
static uint32_t* g_hf_sync[64] = {nullptr};
static uint32_t g_hf_enqueued[64] = {0};
static uint32_t* HfSyncWords(int* dev_out) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  *dev_out = dev;
  if (!g_hf_sync[dev]) {
    if (hipMalloc((void**)&g_hf_sync[dev], 18 * sizeof(uint32_t)) != hipSuccess) return nullptr;      // [0] unused, [1] last HF launch whose workgroups have all started, [2, 18) a start counter per launch in flight
    (void)hipMemset(g_hf_sync[dev], 0, 18 * sizeof(uint32_t));
  }
  return g_hf_sync[dev];
}
// (1 << 24) / (i + 1), i < 64 (context_predict.h kDivLookup), in device memory: the SIMT LF kernel's weighted-predictor lanes read it through the vector L1
static const uint32_t* WpDivTable() {
  static std::mutex mu;
  static uint32_t* table[64] = {nullptr};
  int dev = 0;
  (void)hipGetDevice(&dev);
  if (dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (!table[dev]) {
    uint32_t host[64];
    for (uint32_t i = 0; i < 64; i++) host[i] = (1u << 24) / (i + 1);
    if (hipMalloc((void**)&table[dev], sizeof(host)) != hipSuccess) return nullptr;
    (void)hipMemcpy(table[dev], host, sizeof(host), hipMemcpyHostToDevice);
  }
  return table[dev];
}
void LaunchLfDecode(const FrameDev* frames, int nframes, int max_lf_groups, const LaunchCfg& cfg, void* stream, const LfSimtPlan* simt) {
  static const bool time_it = getenv("JXL_HIP_TIME_LF") != nullptr;     // experiments: blocking per-kernel times on stderr
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  if (time_it) { for (auto& e : ev) (void)hipEventCreate(&e); (void)hipEventRecord(ev[0], (hipStream_t)stream); }
  auto place = [&]() {
    // varblock placement: band starts (prefix sums), the walk of every band (one per lane), records -> block info / varblock lists
    if (time_it) (void)hipEventRecord(ev[1], (hipStream_t)stream);
    hipLaunchKernelGGL(LfBandStartKernel, dim3(DivUp(max_lf_groups, 4), nframes), dim3(256), 0, (hipStream_t)stream, frames);
    if (simt && simt->num_units) {
      // (a handful of frames: one band per wavefront — the lane-per-band walk waits for a global load per varblock, 7 ms for a 4K frame against 1)
      static const bool no_wave_place = getenv("JXL_HIP_NO_WAVE_PLACE") != nullptr;
      if (simt->num_units <= 512 && !no_wave_place) hipLaunchKernelGGL(LfPlaceWaveKernel, dim3(simt->num_units), dim3(64), 0, (hipStream_t)stream, frames, simt->units, simt->num_units);
      else hipLaunchKernelGGL(LfPlaceSimtKernel, dim3(DivUp((int)simt->num_units, (int)kPlaceLanes)), dim3(64), 0, (hipStream_t)stream, frames, simt->units, simt->num_units);
      hipLaunchKernelGGL(LfPlaceExpandKernel, dim3(simt->num_units), dim3(256), 0, (hipStream_t)stream, frames, simt->units, simt->num_units);
    }
    if (time_it) {
      (void)hipEventRecord(ev[2], (hipStream_t)stream); (void)hipEventSynchronize(ev[2]);
      float a = 0, b = 0; (void)hipEventElapsedTime(&a, ev[0], ev[1]); (void)hipEventElapsedTime(&b, ev[1], ev[2]);
      fprintf(stderr, "[jxl-hip] LF stage: %d frames, SIMT %u lanes / %u per wavefront: decode %.2f ms, placement %.2f ms\n", nframes, simt ? simt->num_lanes : 0u, simt ? simt->lanes_per_wave : 0u, a, b);
      for (auto& e : ev) (void)hipEventDestroy(e);
    }
  };
  // this launch: one wavefront per stream for every frame (shorter latency, the whole GPU's SIMDs) — not for batches whose SIMT launch is the weighted-predictor
  // instantiation: per stream the lane-serial form is the faster one there (1.5 against 2.8 us per sample), four of these wide launches at the start of a cold
  // pipeline took 1.0-2.3 s each (profiles/r04_notes.md)
  static const bool wide_wp = getenv("JXL_HIP_WIDE_WP") != nullptr;       // experiments
  static const int force_big_env = getenv("JXL_HIP_LF_FORCE_BIG") ? atoi(getenv("JXL_HIP_LF_FORCE_BIG")) : 0;
  const int wide = cfg.lf_wide_once && (wide_wp || !(simt && simt->num_lanes && simt->any_wp));
  int simt_mode = wide ? 1 : 0;           // LfDecodeKernel's take_simt_frames: 0 legacy frames only, 1 every frame, 2 legacy frames + the streams the SIMT kernel handed back
  if (simt && simt->num_lanes && !wide) {
    // SIMT frames: the entropy decode on a handful of wavefronts (one stream per lane)
    const uint32_t lpw = std::min(64u, std::max(1u, simt->lanes_per_wave));
    static const int lf_prio = getenv("JXL_HIP_LF_PRIO") ? atoi(getenv("JXL_HIP_LF_PRIO")) : 0;
    static const int wpb_env = getenv("JXL_HIP_LF_WAVES_PER_WG") ? atoi(getenv("JXL_HIP_LF_WAVES_PER_WG")) : 1;
    const int nwaves = DivUp((int)simt->num_lanes, (int)lpw), wpb = std::max(1, std::min(4, wpb_env));
    const int lf_flags = lf_prio | (cfg.lf_wp_narrow_test ? 8 : 0);
    const dim3 grid(DivUp(nwaves, wpb)), block(64 * wpb);
    const uint32_t* wp_div = WpDivTable();
    static const bool no_quad = getenv("JXL_HIP_LF_NOQUAD") != nullptr;      // A/B: the one-lane-per-stream weighted-predictor instantiation
    if (simt->any_wp && !no_quad) {
      // four lanes per stream: at most 16 streams per wavefront
      const uint32_t lpq = std::min(16u, lpw);
      const int nwq = DivUp((int)simt->num_lanes, (int)lpq);
      hipLaunchKernelGGL((LfDecodeSimtKernel<true, true, true>), dim3(DivUp(nwq, wpb)), block, 0, (hipStream_t)stream, frames, simt->streams, simt->lanes, simt->luts, simt->num_lanes, lpq, lf_flags, wp_div);
    } else if (simt->any_wp) hipLaunchKernelGGL((LfDecodeSimtKernel<true, true>), grid, block, 0, (hipStream_t)stream, frames, simt->streams, simt->lanes, simt->luts, simt->num_lanes, lpw, lf_flags, wp_div);
    else if (simt->any_general) hipLaunchKernelGGL((LfDecodeSimtKernel<false, true>), grid, block, 0, (hipStream_t)stream, frames, simt->streams, simt->lanes, simt->luts, simt->num_lanes, lpw, lf_flags, wp_div);
    else hipLaunchKernelGGL((LfDecodeSimtKernel<false, false>), grid, block, 0, (hipStream_t)stream, frames, simt->streams, simt->lanes, simt->luts, simt->num_lanes, lpw, lf_flags, wp_div);
    if (!simt->any_legacy && !simt->any_wp) { place(); return; }
    simt_mode = simt->any_wp ? 2 : 0;         // the weighted-predictor lanes may hand streams back: LfDecodeKernel follows for those (and for the legacy frames)
  }
  // dynamic LDS: LUT + scratch + tree copy + as much of the entropy code as needed / the budget allows (right-sized so
  // that several LF groups fit one CU)
  const uint32_t tree_cap = (uint32_t)std::min(cfg.max_tree_nodes, kLdsTreeMax);
  // (the pass that only takes the streams the weighted-predictor SIMT lanes handed back — normally none — keeps the entropy code out of the LDS: its 256 workgroups
  // have to find room on CUs that the HF stage and the pixel kernels fill, and the launch holds up the placement kernels behind it until the last one has been dispatched)
  const bool redo_only_launch = simt_mode == 2 && simt && !simt->any_legacy;
  // (+ the wave-wide decoder's second copy of the alias tables, StageCode(with_wide): 10 bytes per slot — when the plain tables fit their budget and the two together 96 KB)
  static const bool no_wide = getenv("JXL_HIP_NO_WAVE_LF") != nullptr;     // A/B: the lane-0 serial fast path
  const uint32_t wide_bytes = (!no_wide && !redo_only_launch && cfg.mod_code_bytes <= cfg.lds_code_budget && cfg.mod_code_bytes * 9 / 4 + 64 <= 96 * 1024) ? (uint32_t)cfg.mod_code_bytes * 5 / 4 + 48 : 0u;
  // (the one-wavefront-per-stream launches of a cold pipeline / of small jobs: the wide layout alone — StageCode stages what the budget admits, the few readers of the plain layout put the
  // entry together again (LdAliasAt).  These launches run beside the HF stage (80 KB of LDS a workgroup) and the pixel kernels of the jobs before: 40 instead of 57 KB per CU is the
  // difference between one and three IDCT / filter workgroups next to them.  JXL_HIP_LF_WIDE_BOTH: both layouts, as the launches that also serve SIMT hand-backs keep)
  static const bool wide_both = getenv("JXL_HIP_LF_WIDE_BOTH") != nullptr;
  const bool wide_only = wide && wide_bytes && !wide_both;
  const uint32_t lds_tables = kLfDecWaves * kWaveLds + tree_cap * 16 + (redo_only_launch || wide_only ? 0u : (uint32_t)std::min(cfg.lds_code_budget, cfg.mod_code_bytes)) + wide_bytes;
  // trees with the weighted predictor: its state rows (channels up to 256 wide — all but the block-info rows) in LDS, one slot per wavefront
  const uint32_t wp_base = cfg.any_wp ? (lds_tables + 15) & ~15u : 0u;
  const uint32_t lds_bytes = wp_base ? wp_base + kLfDecWaves * kWpLdsBytes : lds_tables;
  static bool attr_set = false;
  if (!attr_set) {
    SetMaxDynamicLds((const void*)LfDecodeKernel<true>, 160 * 1024 - 2048, "LfDecodeKernel<true>");
    SetMaxDynamicLds((const void*)LfDecodeKernel<false>, 160 * 1024 - 2048, "LfDecodeKernel<false>");
    attr_set = true;
  }
  // batches that fill the GPU: four groups per workgroup (two per wavefront, paired large + small); otherwise one per wavefront
  // (cfg.lf_force_big — tests: 1 the four-groups-per-workgroup shape whatever the batch size, 2 the same under the uncapped instantiation, -1 never)
  const int force_big = cfg.lf_force_big ? cfg.lf_force_big : (wide ? force_big_env : 0);
  const bool big4 = force_big > 0 || (force_big == 0 && (size_t)nframes * DivUp(max_lf_groups, (int)kLfDecGroups) >= 128);
  const bool big = big4 && force_big != 2;
  const uint32_t gpb = big4 ? kLfDecGroups : kLfDecWaves;
  if (big && cfg.lf_head_start) {
    std::lock_guard<std::mutex> lock(g_hf_sync_mu);
    int dev;
    if (uint32_t* sync = HfSyncWords(&dev))
      hipLaunchKernelGGL(HeadStartKernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sync, g_hf_enqueued[dev] + 1, (uint64_t)200000);   // the next HF launch, or 2 ms
  }
  if (big) {      // (the register cap — 170 VGPRs, with spills — leaves room for the wavefronts of other stages; for the first launches of a cold pipeline the uncapped instantiation measured the same: profiles/r05_notes.md)
    hipLaunchKernelGGL(LfDecodeKernel<true>, dim3(DivUp(max_lf_groups, (int)gpb), nframes), dim3(64 * kLfDecWaves), lds_bytes, (hipStream_t)stream, frames, gpb, tree_cap, lds_tables, simt_mode, wp_base);
  } else {
    hipLaunchKernelGGL(LfDecodeKernel<false>, dim3(DivUp(max_lf_groups, (int)gpb), nframes), dim3(64 * kLfDecWaves), lds_bytes, (hipStream_t)stream, frames, gpb, tree_cap, lds_tables, simt_mode, wp_base);
  }
  place();
}
void LaunchLfPost(const FrameDev* frames, int nframes, int max_bw, int max_bh, int max_groups, void* stream) {
  dim3 block(64, 4), grid(DivUp(max_bw, 64), DivUp(max_bh, 4), nframes);
  hipLaunchKernelGGL(LfDequantKernel, grid, block, 0, (hipStream_t)stream, frames);
  hipLaunchKernelGGL(LfSmoothKernel, grid, block, 0, (hipStream_t)stream, frames);
  hipLaunchKernelGGL(LlfSigmaKernel, grid, block, 0, (hipStream_t)stream, frames);
  hipLaunchKernelGGL(LlfKernel, dim3(max_groups, nframes), dim3(256), 0, (hipStream_t)stream, frames);
}
// Every IDCT kernel skips a frame whose status word is set, so nobody puts zeros back where that frame's HF stage wrote before it failed: this kernel does, right
// behind the HF stage — the coefficient planes are clean again for whichever decode uses them next (pipelines rotate a few sets between many batches).
__global__ __launch_bounds__(256) void ZeroFailedCoefKernel(const FrameDev* __restrict__ frames) {
  const FrameDev& f = frames[blockIdx.y];
  if (f.is_modular || !FrameFailed(f)) return;
  const size_t n16 = (size_t)f.num_groups * 65536 / 4;
  for (int c = 0; c < 3; c++) {
    int4* p = reinterpret_cast<int4*>(f.coeff[c]);
    if (!p) continue;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) p[i] = make_int4(0, 0, 0, 0);
  }
}
void LaunchZeroFailedCoefficients(const FrameDev* frames, int nframes, void* stream) {
  if (nframes > 0) hipLaunchKernelGGL(ZeroFailedCoefKernel, dim3(32, nframes), dim3(256), 0, (hipStream_t)stream, frames);
}
void LaunchHfDecode(const FrameDev* frames, int nframes, int max_groups, const LaunchCfg& cfg, void* stream) {
  // latency mode (one stream per wavefront asked for) with plain frames whose AC code fits the LDS in the wide layout: the wave-wide kernel
  static const bool no_wave_hf = getenv("JXL_HIP_NO_WAVE_HF") != nullptr;
  if (!no_wave_hf && cfg.lane_stride_hf == 1 && cfg.hf_lanes_per_wave == 1 && !cfg.any_subsampled && !cfg.any_multipass && !cfg.any_prefix_ac && !cfg.skip_hf && cfg.max_passes == 0) {
    const uint32_t code_bytes = (uint32_t)cfg.ac_code_bytes * 5 / 4 + 64;      // cfg + context map + 10 bytes per alias slot (ac_code_bytes counts 8)
    const uint32_t lds = kHwCodeOff + code_bytes;
    if (lds <= 150 * 1024) {
      static bool attr = false;
      if (!attr) { SetMaxDynamicLds((const void*)HfDecodeWaveKernel, 160 * 1024 - 2048, "HfDecodeWaveKernel"); attr = true; }
      hipLaunchKernelGGL(HfDecodeWaveKernel, dim3(DivUp(max_groups, (int)kHwWaves), nframes), dim3(64 * kHwWaves), lds, (hipStream_t)stream, frames, lds);
      return;
    }
  }
  if (cfg.lane_stride_hf == 1) {   // SIMT: one group stream per lane
#ifndef JXL_HF_WIDE_ALIAS
    const int code_lds = cfg.ac_code_bytes_compact;                    // the all-in-LDS instantiations stage the compact form (StageCodeCompact)
#else
    const int code_lds = cfg.ac_code_bytes;
#endif
    const bool all_lds = code_lds <= cfg.lds_code_budget;              // every frame's AC code fits the LDS budget
    // lanes per workgroup: the groups of a frame spread evenly over as few workgroups as possible, whole wavefronts (a
    // 3840x2160 frame has 135 groups: 192 threads, 136 lane regions, 80 KB of LDS instead of 99 KB — which is what lets two
    // LF workgroups share the CU with it)
    // (at most ~136 group streams per workgroup — a 4K frame's 135 in one —: an 8K frame's 510 spread over four workgroups of 128 lanes, 32 per wavefront, decode in a
    // third of the time two workgroups of 255 took — 64 streams per wavefront run every path of the lock-step loop in every iteration)
    static const int lanes_cap = getenv("JXL_HIP_HF_LANES_PER_WG") ? std::max(16, std::min((int)kSimtMaxThreads, atoi(getenv("JXL_HIP_HF_LANES_PER_WG")))) : 136;
    const int nblk = DivUp(max_groups, cfg.hf_lanes_per_wg > 0 ? std::max(16, std::min((int)kSimtMaxThreads, cfg.hf_lanes_per_wg)) : lanes_cap);
    const uint32_t lanes = (uint32_t)DivUp(max_groups, nblk);
    static const int lpw_env = getenv("JXL_HIP_HF_LANES") ? atoi(getenv("JXL_HIP_HF_LANES")) : 0;
    // streams per wavefront: four wavefronts per workgroup, one per SIMD (two of these on one SIMD slow each other down
    // more than the sparser lanes gain: 5 wavefronts 64 ms, 4: 49 ms, 3: 53 ms per 256 4K frames)
    // (latency mode, cfg.hf_lanes_per_wave: a few dozen frames at most — one to four streams per wavefront: a lane that has its wavefront nearly to itself skips the paths
    // the other lanes of a dense wavefront drag it through: 26 ms for one 4K frame at 1 stream per wavefront, 29.5 ms for 64 frames at 4, against 40-41 ms at the dense packing)
    uint32_t lpw = cfg.hf_lanes_per_wave > 0 ? (uint32_t)std::min(64, cfg.hf_lanes_per_wave) : lpw_env >= 1 && lpw_env <= 64 ? (uint32_t)lpw_env : (uint32_t)DivUp((int)lanes, 4);
    lpw = std::max(lpw, (uint32_t)DivUp((int)lanes, 16));                        // at most 16 wavefronts per workgroup
    const uint32_t threads = (uint32_t)DivUp((int)lanes, (int)lpw) * 64;
    const uint32_t lds_need = kSimtCodeOff + (uint32_t)(all_lds ? code_lds : std::min(cfg.lds_code_budget, cfg.ac_code_bytes)) + (lanes + 1) * kSimtLaneBytes;
    // (JXL_HIP_HF_LDS_MIN, experiments: a floor under the request — above half the CU's 160 KB no two HF workgroups share a CU)
    static const uint32_t lds_floor = getenv("JXL_HIP_HF_LDS_MIN") ? (uint32_t)atoi(getenv("JXL_HIP_HF_LDS_MIN")) : 0u;
    const uint32_t lds = std::min(std::max(lds_need, lds_floor), 160u * 1024 - 2048);
    static bool attr = false;
    if (!attr) {
      SetMaxDynamicLds((const void*)HfDecodeSimtKernel<true, false, false>, 160 * 1024 - 2048, "HfDecodeSimtKernel<true, false, false>");
      SetMaxDynamicLds((const void*)HfDecodeSimtKernel<true, true, true>, 160 * 1024 - 2048, "HfDecodeSimtKernel<true, true, true>");
      SetMaxDynamicLds((const void*)HfDecodeSimtKernel<false, true, true>, 160 * 1024 - 2048, "HfDecodeSimtKernel<false, true, true>");
      attr = true;
    }
    const dim3 grid(nblk, nframes);
    uint32_t* sync = nullptr; uint32_t epoch = 0;
    {
      std::lock_guard<std::mutex> lock(g_hf_sync_mu);
      int dev;
      if ((sync = HfSyncWords(&dev)) != nullptr) epoch = ++g_hf_enqueued[dev];
    }
    // the common case gets its own instantiation: tables in LDS, no chroma subsampling, no progressive passes (every instruction of
    // the lock-step loop is paid by all ~37 000 iterations of a frame); everything else takes the general one
    static const int hf_prio = getenv("JXL_HIP_HF_PRIO") ? atoi(getenv("JXL_HIP_HF_PRIO")) : 1;
    const bool plain = all_lds && !cfg.any_subsampled && !cfg.any_multipass;
    if (plain) hipLaunchKernelGGL((HfDecodeSimtKernel<true, false, false>), grid, dim3(threads), lds, (hipStream_t)stream, frames, lds, lanes, lpw, sync, epoch, hf_prio);
    else if (all_lds) hipLaunchKernelGGL((HfDecodeSimtKernel<true, true, true>), grid, dim3(threads), lds, (hipStream_t)stream, frames, lds, lanes, lpw, sync, epoch, hf_prio);
    else hipLaunchKernelGGL((HfDecodeSimtKernel<false, true, true>), grid, dim3(threads), lds, (hipStream_t)stream, frames, lds, lanes, lpw, sync, epoch, hf_prio);
    if (cfg.any_prefix_ac) {   // frames with a prefix-coded AC stream: one group stream per wavefront lane 0, tables in global memory
      static bool attr2 = false;
      if (!attr2) { SetMaxDynamicLds((const void*)HfDecodeKernel, 160 * 1024 - 2048, "HfDecodeKernel"); attr2 = true; }
      hipLaunchKernelGGL(HfDecodeKernel, dim3(DivUp(max_groups, 512 / 64), nframes), dim3(512), 128, (hipStream_t)stream, frames, 64, 128u, 1);
    }
    return;
  }
  const int threads = cfg.hf_block_threads;
  const int per_block = threads / cfg.lane_stride_hf;
  const uint32_t lds_bytes = 128 + (uint32_t)std::min(cfg.lds_code_budget, cfg.ac_code_bytes);
  static bool attr_set = false;
  if (!attr_set) { SetMaxDynamicLds((const void*)HfDecodeKernel, 160 * 1024 - 2048, "HfDecodeKernel"); attr_set = true; }
  dim3 grid(DivUp(max_groups, per_block), nframes);
  hipLaunchKernelGGL(HfDecodeKernel, grid, dim3(threads), lds_bytes, (hipStream_t)stream, frames, cfg.lane_stride_hf, lds_bytes, 0);
}
// JXL_HIP_DEBUG_SYNC: names the launch that the runtime rejected (a rejected launch leaves an error code behind that the next runtime call of the process would report)
static void DebugLaunch(const char* what) {
  static const bool on = getenv("JXL_HIP_DEBUG_SYNC") != nullptr;
  if (!on) return;
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) fprintf(stderr, "[jxl-hip] launch of %s rejected: %s\n", what, hipGetErrorString(e));
}
void LaunchIdct(const FrameDev* frames, int nframes, int max_groups, int max_bw, int max_bh, const LaunchCfg& cfg, void* stream) {
  DebugLaunch("(something before the IDCT stage)");
  // 64x64 tiles for frames with varblocks beyond 32x32, 32x32 tiles (a quarter of the LDS) for the others; each in a
  // variant with and without the 8x8 special transforms.  Every frame is taken by exactly one of the four.
  const bool all = !cfg.idct_flags_known;
  static const int nocoef = getenv("JXL_HIP_IDCT_NOCOEF") ? 4 : 0;
  {
    const int tiles_x = DivUp(max_bw, 8), tiles_y = DivUp(max_bh, 8);
    const dim3 grid(tiles_x * tiles_y, nframes);
    const size_t lds = 3 * TileGeom<8>::kPlane * sizeof(float);
    if (all || cfg.need_tile8_plain) hipLaunchKernelGGL((IdctTileKernel<8, false>), grid, dim3(256), lds, (hipStream_t)stream, frames, tiles_x, cfg.force_generic_idct);
    if (all || cfg.need_tile8_special) hipLaunchKernelGGL((IdctTileKernel<8, true>), grid, dim3(256), lds, (hipStream_t)stream, frames, tiles_x, cfg.force_generic_idct);
    DebugLaunch("IdctTileKernel<8>");
  }
  {
    const int tiles_x = DivUp(max_bw, 4), tiles_y = DivUp(max_bh, 4);
    const dim3 grid(tiles_x * tiles_y, nframes);
    static const size_t pad = getenv("JXL_HIP_PAD_LDS_IDCT") ? (size_t)atoi(getenv("JXL_HIP_PAD_LDS_IDCT")) : 0;   // (experiments, as JXL_HIP_PAD_LDS_FILTER; up to the 64 KB a launch gets without asking)
    const size_t lds = 3 * TileGeom<4>::kPlane * sizeof(float) + pad;
    if (all || cfg.need_tile4_plain) hipLaunchKernelGGL((IdctTileKernel<4, false>), grid, dim3(JXL_IDCT_T4), lds, (hipStream_t)stream, frames, tiles_x, cfg.force_generic_idct);
    if (all || cfg.need_tile4_special) hipLaunchKernelGGL((IdctTileKernel<4, true>), grid, dim3(JXL_IDCT_T4), lds, (hipStream_t)stream, frames, tiles_x, cfg.force_generic_idct | nocoef);
    DebugLaunch("IdctTileKernel<4>");
  }
  if (cfg.any_subsampled) {
    static const bool per_block = getenv("JXL_HIP_SUBSAMPLED_IDCT_PER_BLOCK") != nullptr;     // A/B: the thread-per-block form
    if (per_block) hipLaunchKernelGGL(IdctSubsampledKernel, dim3(max_groups, nframes), dim3(256), 0, (hipStream_t)stream, frames);
    else hipLaunchKernelGGL(IdctSubsampledTileKernel, dim3(DivUp(max_bw, kSubTileBx) * DivUp(max_bh, kSubTileBy), 3, nframes), dim3(256), 0, (hipStream_t)stream, frames);
  }
  DebugLaunch("IdctSubsampledKernel");
  if (cfg.force_generic_idct || !cfg.idct_flags_known || cfg.any_irregular_blocks)
    hipLaunchKernelGGL(IdctKernel, dim3(max_groups, nframes), dim3(256), 0, (hipStream_t)stream, frames, cfg.force_generic_idct);   // irregular frames only
  if (!cfg.idct_flags_known || cfg.any_big_blocks)
    hipLaunchKernelGGL(BigIdctKernel, dim3(max_groups, nframes), dim3(256), kBigLds, (hipStream_t)stream, frames);                  // frames with DCT128/256 varblocks only
  DebugLaunch("IdctKernel / BigIdctKernel");
}
void LaunchFilters(const FrameDev* frames, int nframes, int max_w, int max_h, const FilterPlan& fp, const LaunchCfg& cfg, void* stream) {
  dim3 block(64, 4), grid(DivUp(max_w, 64), DivUp(max_h, 4), nframes);
  const int unfused = cfg.force_unfused_filters;
  if (fp.any_fused && !unfused) {
    static bool attr = false;
    // (JXL_HIP_PAD_LDS_FILTER, experiments: bytes of LDS the launch asks for beyond what the kernel uses — occupancy as beside an HF workgroup, on an idle GPU)
    static const size_t pad = getenv("JXL_HIP_PAD_LDS_FILTER") ? (size_t)atoi(getenv("JXL_HIP_PAD_LDS_FILTER")) : 0;
    if (!attr) { SetMaxDynamicLds((const void*)FusedGabEpf1OutKernel, (int)(kFusedLds + pad), "FusedGabEpf1OutKernel"); attr = true; }
    static const int swizzle = getenv("JXL_HIP_NO_XCD_SWIZZLE") ? 0 : 1;
    const int tiles_x = DivUp(max_w, kFtW);
    hipLaunchKernelGGL(FusedGabEpf1OutKernel, dim3(tiles_x * DivUp(max_h, kFtH), 1, nframes), dim3(256), kFusedLds + pad, (hipStream_t)stream, frames, unfused, tiles_x, swizzle);
  }
  if (!fp.any_unfused && !unfused) return;
  // (cfg.debug_stop_after, testing: 2 = stop after gaborish, 3 / 4 / 5 = after EPF pass 0 / 1 / 2 — the planes are then read back, JxlHipBatchDebugRead)
  const int stop = cfg.debug_stop_after ? cfg.debug_stop_after : 99;
  // (fuse_out bit 0: the last EPF pass writes the pixels; bit 1: gaborish inside the first EPF pass of those frames — JXL_HIP_NO_GAB_FOLD: the separate pass, an A/B knob)
  static const bool no_gab_fold = getenv("JXL_HIP_NO_GAB_FOLD") != nullptr;
  static const bool no_p12 = getenv("JXL_HIP_NO_EPF12") != nullptr;           // (A/B knob: passes 1 and 2 as kernels of their own)
  const int fuse_out = cfg.debug_stop_after ? 0 : ((no_gab_fold ? 1 : 3) | (no_p12 ? 0 : 4));
  if (fp.any_gab) hipLaunchKernelGGL(GaborishKernel, grid, block, 0, (hipStream_t)stream, frames, unfused, fuse_out);
  // the EPF passes on LDS tiles (the per-pixel kernels they replaced in round 5 — EpfKernel<PASS>, 90 spilled registers in the first pass — are in the history)
  const int etx = DivUp(max_w, kEtT);
  const dim3 tgrid(etx * DivUp(max_h, kEtT), 1, nframes);
  if (fp.max_epf >= 3 && stop >= 3) hipLaunchKernelGGL(EpfTileKernel<0>, tgrid, dim3(256), 0, (hipStream_t)stream, frames, unfused, etx, fuse_out);
  if (fp.max_epf >= 1 && stop >= 4) hipLaunchKernelGGL(EpfTileKernel<1>, tgrid, dim3(256), 0, (hipStream_t)stream, frames, unfused, etx, fuse_out);
  if (fp.max_epf >= 2 && stop >= 5) hipLaunchKernelGGL(EpfTileKernel<2>, tgrid, dim3(256), 0, (hipStream_t)stream, frames, unfused, etx, fuse_out);
  if (fp.max_epf >= 2 && (fuse_out & 4)) hipLaunchKernelGGL(EpfTile12Kernel, tgrid, dim3(256), 0, (hipStream_t)stream, frames, unfused, etx, fuse_out);
}
void LaunchOutput(const FrameDev* frames, int nframes, int max_w, int max_h, const FilterPlan& fp, const LaunchCfg& cfg, void* stream) {
  if (!fp.any_unfused && !cfg.force_unfused_filters) return;
  const int ow = std::max(max_w, fp.max_out_w), oh = std::max(max_h, fp.max_out_h);   // upsampled frames write more pixels than they code
  dim3 block(64, 4), grid(DivUp(ow, 64), DivUp(oh, 4), nframes);
  if (fp.any_upsampled) hipLaunchKernelGGL(UpsampleKernel, grid, block, 0, (hipStream_t)stream, frames);
  hipLaunchKernelGGL(OutputKernel, grid, block, 0, (hipStream_t)stream, frames, cfg.force_unfused_filters, cfg.debug_stop_after ? 0 : 1);
}
void LaunchModularGlobal(const FrameDev* frames, int nframes, const LaunchCfg& cfg, void* stream);
// LDS plan of the Modular kernels: per-wavefront regions, tree region (whole tree or pruned per-wavefront slices), the
// entropy code, and — for weighted-predictor trees — one WP-state slot per wavefront.  Everything should be LDS-resident
// (a lookup in global memory costs ~10x); when it does not all fit, the tree region shrinks first (large trees are pruned
// per stream anyway), then the code falls back to the shared budget.
static void PlanModularLds(const LaunchCfg& cfg, uint32_t* nwaves_io, uint32_t* tree_cap, uint32_t* lds_tables, uint32_t* wp_base, uint32_t* lds_total) {
  // (the kernels' static __shared__ arrays — 5 KB in ModularGroupFastKernel — come on top of the dynamic part: a plan that ends above 160 KB in total is rejected by the
  // runtime at launch.  Round 4: the limit was 156 KB of dynamic LDS, which a stream with large alias tables reached — the launch was refused and left its error code behind)
  const uint32_t limit = 160 * 1024 - 8192;
  uint32_t nwaves = *nwaves_io;
  const bool pruned = cfg.max_tree_nodes > kLdsTreeMax;          // the tree is copied per wavefront, reduced to what its stream can reach
  uint32_t cap = (uint32_t)std::min(cfg.max_tree_nodes, kLdsTreeMax);
  if (pruned && nwaves > 1) { nwaves = 2; cap = 2 * kLdsTreeMax; }   // two wavefronts, 1024 nodes each
  uint32_t code = (uint32_t)std::min(cfg.mod_code_bytes, 96 * 1024);
  auto wp = [&]() { return cfg.any_wp ? nwaves * kWpLdsBytes : 0u; };
  auto total = [&]() { return nwaves * kWaveLds + cap * 16 + code + 16 + wp(); };
  while (total() > limit && pruned && cap > 512) cap /= 2;
  if (total() > limit) code = (uint32_t)std::min(cfg.lds_code_budget, cfg.mod_code_bytes);
  // the wave-wide decoders' second copy of the alias tables (StageCode with_wide: 10 bytes per slot), when the plain tables are staged whole and both fit
  static const bool no_wide = getenv("JXL_HIP_NO_WAVE_LF") != nullptr;
  if (!no_wide && code == (uint32_t)cfg.mod_code_bytes) {
    const uint32_t wide = code * 5 / 4 + 64;
    if (total() + wide <= limit) code += wide;               // both layouts
    else { const uint32_t plain = code; code = wide; if (total() > limit) code = plain; }     // the wide one alone (bench.jxl: 66 KB of plain tables), or the plain one as before
  }
  *nwaves_io = nwaves;
  *tree_cap = cap;
  *lds_tables = nwaves * kWaveLds + cap * 16 + code;
  if (wp() && total() <= limit) { *wp_base = (*lds_tables + 15) & ~15u; *lds_total = *wp_base + wp(); }
  else { *wp_base = 0; *lds_total = *lds_tables; }
}
void LaunchModularGroups(const FrameDev* frames, int nframes, int max_units, const LaunchCfg& cfg, void* stream) {
  const int max_lf_groups = max_units, max_groups = 0;
  uint32_t nwaves = kLfWaves, tree_cap, lds_tables, wp_base, lds_bytes;
  PlanModularLds(cfg, &nwaves, &tree_cap, &lds_tables, &wp_base, &lds_bytes);
  static bool attr_set = false;
  if (!attr_set) { SetMaxDynamicLds((const void*)ModularGroupFastKernel<false>, 160 * 1024 - 8192, "ModularGroupFastKernel<false>"); SetMaxDynamicLds((const void*)ModularGroupFastKernel<true>, 160 * 1024 - 8192, "ModularGroupFastKernel<true>"); attr_set = true; }
  const bool bigtree = cfg.max_tree_nodes > kBigTreeNodes;
  if (bigtree) hipLaunchKernelGGL(ModularGroupFastKernel<true>, dim3(DivUp(max_lf_groups + max_groups, (int)nwaves), nframes), dim3(64 * nwaves), lds_bytes, (hipStream_t)stream, frames, tree_cap, lds_tables, wp_base, 0u);
  else hipLaunchKernelGGL(ModularGroupFastKernel<false>, dim3(DivUp(max_lf_groups + max_groups, (int)nwaves), nframes), dim3(64 * nwaves), lds_bytes, (hipStream_t)stream, frames, tree_cap, lds_tables, wp_base, 0u);
  if (getenv("JXL_HIP_DEBUG_SYNC")) fprintf(stderr, "[jxl-hip] ModularGroupFastKernel: grid %d x %d, %u threads, %u B of dynamic LDS (tree cap %u, tables %u, wp base %u; max nodes %d, code %d B, any_wp %d)\n",
                                            DivUp(max_lf_groups + max_groups, (int)nwaves), nframes, 64 * nwaves, lds_bytes, tree_cap, lds_tables, wp_base, cfg.max_tree_nodes, cfg.mod_code_bytes, cfg.any_wp);
  if (cfg.any_local_trees) {   // units with a tree / code of their own: one wavefront per workgroup, each staging its own tables
    nwaves = 1;
    PlanModularLds(cfg, &nwaves, &tree_cap, &lds_tables, &wp_base, &lds_bytes);
    if (bigtree) hipLaunchKernelGGL(ModularGroupFastKernel<true>, dim3(max_lf_groups + max_groups, nframes), dim3(64), lds_bytes, (hipStream_t)stream, frames, tree_cap, lds_tables, wp_base, 1u);
    else hipLaunchKernelGGL(ModularGroupFastKernel<false>, dim3(max_lf_groups + max_groups, nframes), dim3(64), lds_bytes, (hipStream_t)stream, frames, tree_cap, lds_tables, wp_base, 1u);
  }
}
uint32_t ModularGroupLdsBytes(const LaunchCfg& cfg) {   // dynamic LDS of the shared-tree launch of ModularGroupFastKernel (tests: plans above the 64 KB default)
  uint32_t nwaves = kLfWaves, tree_cap, lds_tables, wp_base, lds_bytes;
  PlanModularLds(cfg, &nwaves, &tree_cap, &lds_tables, &wp_base, &lds_bytes);
  return lds_bytes;
}
void LaunchModularGlobal(const FrameDev* frames, int nframes, const LaunchCfg& cfg, void* stream) {
  uint32_t nwaves = 1, tree_cap, lds_tables, wp_base, lds_bytes;
  PlanModularLds(cfg, &nwaves, &tree_cap, &lds_tables, &wp_base, &lds_bytes);
  static bool attr_set = false;
  if (!attr_set) { SetMaxDynamicLds((const void*)ModularGlobalFastKernel, 160 * 1024 - 8192, "ModularGlobalFastKernel"); attr_set = true; }
  hipLaunchKernelGGL(ModularGlobalFastKernel, dim3(nframes), dim3(64), lds_bytes, (hipStream_t)stream, frames, tree_cap, lds_tables, wp_base);
}
void LaunchModInvSqueezeBatch(const SqueezeBatch& b, int n, int horizontal, uint32_t aw, uint32_t ah, uint32_t rw, uint32_t rh, void* stream) {
  if (n <= 0) return;
  if (horizontal) { if (ah) hipLaunchKernelGGL(ModInvSqueezeHKernel, dim3((ah + 63) / 64, n), dim3(64), 0, (hipStream_t)stream, b, aw, rw, ah); }
  else if (aw) hipLaunchKernelGGL(ModInvSqueezeVKernel, dim3((aw + 255) / 256, n), dim3(256), 0, (hipStream_t)stream, b, aw, ah, rh);
}
void LaunchModInvSqueeze(const int32_t* avg, const int32_t* res, int32_t* out, int horizontal, uint32_t aw, uint32_t ah, uint32_t rw, uint32_t rh, void* stream) {
  SqueezeBatch b;
  memset(&b, 0, sizeof(b));
  b.avg[0] = avg; b.res[0] = res; b.out[0] = out;
  LaunchModInvSqueezeBatch(b, 1, horizontal, aw, ah, rw, rh, stream);
}
void LaunchModRct(int32_t* a, int32_t* b, int32_t* c, size_t n, uint32_t rct_type, void* stream) {
  hipLaunchKernelGGL(ModRctKernel, dim3((unsigned)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, c, n, rct_type);
}
void LaunchModPalette(const int32_t* pal, int32_t* const* out, uint32_t nb_colors, uint32_t num_c, uint32_t bit_depth, size_t n, void* stream) {
  ModPaletteArgs a;
  a.pal = pal; a.nb_colors = nb_colors; a.num_c = num_c; a.bit_depth = bit_depth; a.n = n;
  for (uint32_t c = 0; c < 4; c++) a.out[c] = c < num_c ? out[c] : nullptr;
  hipLaunchKernelGGL(ModPaletteKernel, dim3((unsigned)std::min<size_t>(4096, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a);
}
void LaunchModPaletteDelta(const int32_t* pal, int32_t* const* out, uint32_t nb_colors, uint32_t num_c, uint32_t bit_depth, uint32_t nb_deltas, uint32_t predictor,
                           uint32_t w, uint32_t h, const WPHeader& wp, int32_t* wp_scratch, uint32_t wp_stride, void* stream) {
  ModPaletteDeltaArgs a;
  a.pal = pal; a.nb_colors = nb_colors; a.num_c = num_c; a.bit_depth = bit_depth; a.nb_deltas = nb_deltas; a.predictor = predictor; a.w = w; a.h = h;
  a.wp = wp; a.wp_scratch = wp_scratch; a.wp_stride = wp_stride;
  for (uint32_t c = 0; c < 4; c++) a.out[c] = c < num_c ? out[c] : nullptr;
  hipLaunchKernelGGL(ModPaletteDeltaKernel, dim3(1), dim3(64), 0, (hipStream_t)stream, a);
}
void LaunchModOutput(const FrameDev* frames, int fidx, const ModOutputArgs& a, int w, int h, void* stream) {
  dim3 block(64, 4), grid(DivUp(w, 64), DivUp(h, 4));
  hipLaunchKernelGGL(ModularOutputKernel, grid, block, 0, (hipStream_t)stream, frames, fidx, a);
}

}  // namespace jxlhip
