// jxl-hip: device-side decode core (entropy decode of JPEG XL sub-streams), shared by HIP kernels and by the host
// header parser (global sections only).  Plain per-thread functions — no wave intrinsics — so the launch layer decides
// how many streams a wave carries (lane stride).  Replaces the per-section work libjxl does under
// JxlDecoderProcessInput (jpegxl-rs/src/decode.rs:238): dec_ans.h (ANS/hybrid-uint), modular/encoding/encoding.cc
// (MA-tree sample decode), dec_group.cc (coefficient decode).  Format facts: SURVEY.md App. B.4-B.6.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>
#include <string.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define JXL_HD __host__ __device__ __forceinline__
#define JXL_HD_NOINLINE __host__ __device__ inline
#else
#define JXL_HD inline
#define JXL_HD_NOINLINE inline
#endif

namespace jxlhip {

// error bits written by kernels into a per-frame status word
enum DevError : uint32_t {
  kErrNone = 0,
  kErrAnsFinalState = 1u << 0,
  kErrOverrun = 1u << 1,
  kErrBadValue = 1u << 2,
  kErrUnsupported = 1u << 3,
  kErrNzeros = 1u << 4,
  kErrVarblock = 1u << 5,
};

// ---- bit reader over 32-bit aligned words -------------------------------------------------------------------------------
struct BitReader {
  const uint32_t* words;
  uint32_t wpos, wend;
  uint64_t buf;
  int avail;
  // data: 4-byte aligned base; bit_pos: absolute bit offset from base; byte_end: end of readable bytes (from base)
  JXL_HD void Init(const uint8_t* base, uint64_t bit_pos, uint64_t byte_end) {
    words = reinterpret_cast<const uint32_t*>(base);
    wpos = (uint32_t)(bit_pos >> 5);
    wend = (uint32_t)((byte_end + 3) >> 2);
    buf = 0; avail = 0;
    Refill();
    int skip = (int)(bit_pos & 31);
    buf >>= skip; avail -= skip;
    Refill();
  }
  JXL_HD void Refill() {
    if (avail <= 32) {
      uint32_t w = wpos < wend ? words[wpos] : 0u;
      wpos++;
      buf |= (uint64_t)w << avail;
      avail += 32;
    }
  }
  JXL_HD uint32_t Peek(int n) const { return (uint32_t)(buf & ((1ull << n) - 1)); }  // n <= 32, after Refill
  JXL_HD void Consume(int n) { buf >>= n; avail -= n; }
  JXL_HD uint32_t Read(int n) {  // n <= 32
    Refill();
    uint32_t v = (uint32_t)(buf & ((1ull << n) - 1));
    buf >>= n; avail -= n;
    return v;
  }
  JXL_HD uint64_t BitPos() const { return (uint64_t)wpos * 32 - (uint64_t)avail; }
};

// ---- entropy code tables -----------------------------------------------------------------------------------------------
// alias entry packed in 64 bits: cutoff[0:8) right[8:16) freq0[16:29) offs1[29:42) freq1[42:55)
JXL_HD uint64_t PackAlias(uint32_t cutoff, uint32_t right, uint32_t freq0, uint32_t offs1, uint32_t freq1) {
  return (uint64_t)cutoff | ((uint64_t)right << 8) | ((uint64_t)freq0 << 16) | ((uint64_t)offs1 << 29) | ((uint64_t)freq1 << 42);
}

struct DevCode {
  const uint8_t* ctx_map;   // num_ctx entries
  const uint32_t* cfg;      // per cluster: split_exponent | msb<<8 | lsb<<16
  const uint64_t* alias;    // [cluster << log_alpha]   (ANS)
  // prefix codes (canonical, bit-serial): per cluster 16 counts + symbol list
  const uint16_t* pfx_count;   // [cluster*16 + len]
  const uint32_t* pfx_sym_off; // [cluster] offset into pfx_syms
  const uint16_t* pfx_syms;
  uint32_t num_ctx, num_clusters, log_alpha, use_prefix;
  // LZ77 (dec_ans.h): tokens >= lz_min_symbol start a copy; ctx_map then holds one more entry (the distance context, last)
  uint32_t lz77, lz_min_symbol, lz_min_length, lz_len_cfg;
};

struct AnsReader {
  uint32_t state;
  JXL_HD void Init(BitReader& br, const DevCode& code) { state = code.use_prefix ? 0x130000u : br.Read(32); }
  JXL_HD bool FinalOk(const DevCode& code) const { return code.use_prefix || state == 0x130000u; }
};

template <typename BR> JXL_HD uint32_t ReadSymbol(BR& br, AnsReader& ans, const DevCode& code, uint32_t cluster) {
  if (code.use_prefix) {
    const uint16_t* cnt = code.pfx_count + cluster * 16;
    const uint16_t* syms = code.pfx_syms + code.pfx_sym_off[cluster];
    if (cnt[0]) return syms[0];  // cnt[0] != 0 flags a zero-bit single-symbol code
    uint32_t c = 0, first = 0, index = 0;
    for (int len = 1; len <= 15; len++) {
      c |= br.Read(1);
      uint32_t count = cnt[len];
      if (c - first < count) return syms[index + (c - first)];
      index += count; first += count;
      first <<= 1; c <<= 1;
    }
    return 0;
  }
  const uint32_t la = code.log_alpha;
  const uint32_t res = ans.state & 0xFFF;
  const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
  const uint64_t e = code.alias[(cluster << la) + i];
  const uint32_t cutoff = (uint32_t)(e & 0xFF), right = (uint32_t)((e >> 8) & 0xFF);
  const uint32_t freq0 = (uint32_t)((e >> 16) & 0x1FFF), offs1 = (uint32_t)((e >> 29) & 0x1FFF), freq1 = (uint32_t)((e >> 42) & 0x1FFF);
  const bool hit = pos >= cutoff;
  const uint32_t sym = hit ? right : i;
  const uint32_t off = hit ? offs1 + pos : pos;
  const uint32_t freq = hit ? freq1 : freq0;
  ans.state = freq * (ans.state >> 12) + off;
  if (ans.state < (1u << 16)) ans.state = (ans.state << 16) | br.Read(16);
  return sym;
}

JXL_HD uint32_t ReadHybridUint(BitReader& br, AnsReader& ans, const DevCode& code, uint32_t ctx) {
  const uint32_t cluster = code.ctx_map[ctx];
  uint32_t tok = ReadSymbol(br, ans, code, cluster);
  const uint32_t cfg = code.cfg[cluster];
  const uint32_t split_exp = cfg & 0xFF, msb = (cfg >> 8) & 0xFF, lsb = (cfg >> 16) & 0xFF;
  const uint32_t split = 1u << split_exp;
  if (tok < split) return tok;
  uint32_t nbits = split_exp - (msb + lsb) + ((tok - split) >> (msb + lsb));
  nbits &= 31;  // (nbits > 32 is rejected by the host for the configs it can see; clamp to stay in range)
  const uint32_t low = tok & ((1u << lsb) - 1);
  tok >>= lsb;
  const uint32_t bits = nbits ? br.Read((int)nbits) : 0;
  const uint32_t hi = (1u << msb) | (tok & ((1u << msb) - 1));
  return (((hi << nbits) | bits) << lsb) | low;
}

JXL_HD int32_t UnpackSigned(uint32_t u) { return (int32_t)((u >> 1) ^ (~(u & 1) + 1)); }

// ---- LZ77 over decoded values (dec_ans.h ANSSymbolReader::ReadHybridUintClustered with lz77 enabled) ------------------------
// window: 2^20 entries of scratch owned by the stream (only positions < num_decoded are ever read)
template <typename BR> JXL_HD uint32_t HybridFromToken(BR& br, uint32_t cfg, uint32_t tok) {
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
struct Lz77State {
  uint32_t* window;          // wmask + 1 entries
  uint32_t num_to_copy, copy_pos, num_decoded, dist_multiplier;
  uint32_t wmask;            // window size - 1: kWindow - 1 (libjxl's 2^20) unless the stream is known to be shorter than a smaller power of two
  static constexpr uint32_t kWindow = 1u << 20, kMask = kWindow - 1;
  JXL_HD void Init(uint32_t* w, uint32_t dist_mult, uint32_t entries = kWindow) { window = w; num_to_copy = copy_pos = num_decoded = 0; dist_multiplier = dist_mult; wmask = entries - 1; }
};
// AC coefficient streams (one per pass and group): at most 3 x (1024 + 65536) values, so a window of 2^18 entries never wraps and behaves
// like libjxl's 2^20-entry one (distances are clamped to the number of decoded values)
constexpr uint32_t kAcLzWindow = 1u << 18;
JXL_HD int32_t Lz77SpecialDistance(uint32_t i, uint32_t mult) {   // kSpecialDistances[i][0] + mult * kSpecialDistances[i][1]
  // the 120 WebP-lossless-style neighbour offsets (dx in -7..8, dy in 0..7), packed as (dx + 7) | dy << 4
  const uint8_t t[120] = {
      0x17, 0x08, 0x18, 0x16, 0x27, 0x09, 0x28, 0x26, 0x19, 0x15, 0x29, 0x25, 0x37, 0x0A, 0x38, 0x36, 0x1A, 0x14, 0x39, 0x35,
      0x2A, 0x24, 0x47, 0x0B, 0x48, 0x46, 0x1B, 0x13, 0x3A, 0x34, 0x49, 0x45, 0x2B, 0x23, 0x57, 0x4A, 0x44, 0x3B, 0x33, 0x0C,
      0x58, 0x56, 0x1C, 0x12, 0x59, 0x55, 0x2C, 0x22, 0x4B, 0x43, 0x5A, 0x54, 0x3C, 0x32, 0x67, 0x0D, 0x68, 0x66, 0x1D, 0x11,
      0x69, 0x65, 0x2D, 0x21, 0x5B, 0x53, 0x4C, 0x42, 0x6A, 0x64, 0x3D, 0x31, 0x77, 0x0E, 0x78, 0x76, 0x5C, 0x52, 0x1E, 0x10,
      0x6B, 0x63, 0x4D, 0x41, 0x79, 0x75, 0x2E, 0x20, 0x7A, 0x74, 0x3E, 0x30, 0x6C, 0x62, 0x5D, 0x51, 0x0F, 0x7B, 0x73, 0x4E,
      0x40, 0x1F, 0x2F, 0x6D, 0x61, 0x3F, 0x7C, 0x72, 0x5E, 0x50, 0x4F, 0x7D, 0x71, 0x6E, 0x60, 0x5F, 0x7E, 0x70, 0x6F, 0x7F,
  };
  const int first = (int)(t[i] & 15) - 7, second = (int)(t[i] >> 4);
  return first + (int32_t)mult * second;
}
// One value of an LZ77-coded stream: `cluster_of(ctx)` / `read_symbol(cluster)` are supplied by the caller (alias tables or
// prefix codes, wherever they live).
template <typename BR, typename ClusterFn, typename SymbolFn, typename CfgFn>
JXL_HD uint32_t Lz77Read(BR& br, Lz77State& lz, uint32_t ctx, uint32_t dist_ctx, uint32_t min_symbol, uint32_t min_length, uint32_t len_cfg,
                         ClusterFn cluster_of, SymbolFn read_symbol, CfgFn cfg_of) {   // cluster_of maps the two "contexts" the caller passes to clusters
  for (;;) {
    if (lz.num_to_copy > 0) {
      const uint32_t v = lz.window[(lz.copy_pos++) & lz.wmask];
      lz.num_to_copy--;
      lz.window[(lz.num_decoded++) & lz.wmask] = v;
      return v;
    }
    const uint32_t cl = cluster_of(ctx);
    const uint32_t tok = read_symbol(cl);
    if (tok >= min_symbol) {
      lz.num_to_copy = HybridFromToken(br, len_cfg, tok - min_symbol) + min_length;
      const uint32_t dcl = cluster_of(dist_ctx);
      const uint32_t dtok = read_symbol(dcl);
      uint32_t distance = HybridFromToken(br, cfg_of(dcl), dtok);
      const uint32_t nspecial = lz.dist_multiplier == 0 ? 0u : 120u;
      if (distance < nspecial) { const int32_t d = Lz77SpecialDistance(distance, lz.dist_multiplier); distance = d < 1 ? 1u : (uint32_t)d; }
      else distance = distance + 1 - nspecial;
      if (distance > lz.num_decoded) distance = lz.num_decoded;
      if (distance > lz.wmask) distance = lz.wmask + 1;
      lz.copy_pos = lz.num_decoded - distance;
      if (distance == 0) { const uint32_t n = lz.num_to_copy <= lz.wmask ? lz.num_to_copy : lz.wmask + 1; for (uint32_t i = 0; i < n; i++) lz.window[i] = 0; }
      if (lz.num_to_copy < min_length) return 0;   // (length overflow: libjxl bails out with 0)
      continue;
    }
    const uint32_t v = HybridFromToken(br, cfg_of(cl), tok);
    lz.window[(lz.num_decoded++) & lz.wmask] = v;
    return v;
  }
}

// ---- MA tree -----------------------------------------------------------------------------------------------------------
// inner node: prop >= 0, val = split value, a = left child (prop > val), b = right child
// leaf:       prop = -1, val = offset, a = predictor | ctx << 8, b = multiplier
struct TreeNode { int32_t prop; int32_t val; uint32_t a; uint32_t b; };

struct WPHeader { int32_t p1, p2, p3[5], w[4]; };

struct ChannelDesc {
  int32_t* data;   // top-left sample
  int32_t w, h;
  int32_t stride;  // in samples
  int32_t hs, vs;           // subsampling shifts (only compared: previous-channel properties need channels of identical geometry)
};

// Previous channels of the same sub-stream with the geometry of the one being decoded, nearest first (context_predict.h
// PrecomputeReferences): what properties 16 + 4r .. 19 + 4r look at.
constexpr int kMaxModRefs = 8;
struct ModRefs { int32_t n; int32_t stride[kMaxModRefs]; const int32_t* data[kMaxModRefs]; };

JXL_HD int FloorLog2u64(uint64_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return 63 - __clzll((long long)x);
#else
  return 63 - __builtin_clzll(x);
#endif
}
JXL_HD int64_t Abs64(int64_t v) { return v < 0 ? -v : v; }
JXL_HD int64_t Min64(int64_t a, int64_t b) { return a < b ? a : b; }
JXL_HD int64_t Max64(int64_t a, int64_t b) { return a > b ? a : b; }

// self-correcting weighted predictor; scratch layout: err[(w+2)*2] then pred_errors[4][(w+2)*2]
struct WPState {
  int32_t* error;
  int32_t* pe[4];
  int32_t xsize;
  int64_t prediction[4];
  int64_t pred;
  JXL_HD void Init(int32_t* scratch, int32_t xs) {
    xsize = xs;
    const int32_t n = (xs + 2) * 2;
    error = scratch;
    for (int i = 0; i < 4; i++) pe[i] = scratch + n * (1 + i);
    for (int32_t i = 0; i < 5 * n; i++) scratch[i] = 0;
  }
  static JXL_HD uint32_t DivLookup(uint32_t i) { return (1u << 24) / (i + 1); }
  static JXL_HD uint32_t ErrorWeight(uint64_t x, uint32_t maxweight) {
    int shift = FloorLog2u64(x + 1) - 5;
    if (shift < 0) shift = 0;
    return 4 + ((maxweight * DivLookup((uint32_t)(x >> shift))) >> shift);
  }
  JXL_HD int64_t Predict(const WPHeader& hdr, int x, int y, int64_t N, int64_t W, int64_t NE, int64_t NW, int64_t NN, int32_t* max_err) {
    const int32_t cur_row = (y & 1) ? 0 : (xsize + 2);
    const int32_t prev_row = (y & 1) ? (xsize + 2) : 0;
    const int32_t pos_N = prev_row + x;
    const int32_t pos_NE = x < xsize - 1 ? pos_N + 1 : pos_N;
    const int32_t pos_NW = x > 0 ? pos_N - 1 : pos_N;
    uint32_t weights[4];
    for (int i = 0; i < 4; i++)
      weights[i] = ErrorWeight((uint64_t)(uint32_t)pe[i][pos_N] + (uint32_t)pe[i][pos_NE] + (uint32_t)pe[i][pos_NW], (uint32_t)hdr.w[i]);
    N *= 8; W *= 8; NE *= 8; NW *= 8; NN *= 8;
    const int64_t teW = x == 0 ? 0 : error[cur_row + x - 1];
    const int64_t teN = error[pos_N];
    const int64_t teNW = error[pos_NW];
    const int64_t sumWN = teN + teW;
    const int64_t teNE = error[pos_NE];
    int64_t p = teW;
    if (Abs64(teN) > Abs64(p)) p = teN;
    if (Abs64(teNW) > Abs64(p)) p = teNW;
    if (Abs64(teNE) > Abs64(p)) p = teNE;
    *max_err = (int32_t)p;
    prediction[0] = W + NE - N;
    prediction[1] = N - (((sumWN + teNE) * hdr.p1) >> 5);
    prediction[2] = W - (((sumWN + teNW) * hdr.p2) >> 5);
    prediction[3] = N - ((teNW * hdr.p3[0] + teN * hdr.p3[1] + teNE * hdr.p3[2] + (NN - N) * hdr.p3[3] + (NW - W) * hdr.p3[4]) >> 5);
    uint32_t wsum = weights[0] + weights[1] + weights[2] + weights[3];
    const int lw = FloorLog2u64(wsum);
    wsum = 0;
    for (int i = 0; i < 4; i++) { weights[i] >>= (lw - 4); wsum += weights[i]; }
    int64_t sum = (int64_t)(wsum >> 1) - 1;
    for (int i = 0; i < 4; i++) sum += prediction[i] * (int64_t)weights[i];
    pred = (sum * (int64_t)DivLookup(wsum - 1)) >> 24;
    if (((teN ^ teW) | (teN ^ teNW)) > 0) return pred;
    const int64_t mx = Max64(W, Max64(NE, N)), mn = Min64(W, Min64(NE, N));
    pred = Max64(mn, Min64(mx, pred));
    return pred;
  }
  JXL_HD void Update(int64_t val, int x, int y) {
    const int32_t cur_row = (y & 1) ? 0 : (xsize + 2);
    const int32_t prev_row = (y & 1) ? (xsize + 2) : 0;
    val *= 8;
    error[cur_row + x] = (int32_t)(pred - val);
    for (int i = 0; i < 4; i++) {
      const int32_t err = (int32_t)((Abs64(prediction[i] - val) + 3) >> 3);
      pe[i][cur_row + x] = err;
      pe[i][prev_row + x + 1] += err;
    }
  }
};

JXL_HD int64_t ClampedGradient(int64_t n, int64_t w, int64_t l) {
  const int64_t m = Min64(n, w), M = Max64(n, w);
  return l < m ? M : (l > M ? m : n + w - l);
}

struct ModularCtx {
  const TreeNode* tree;
  const DevCode* code;
  WPHeader wp;
  uint32_t uses_wp;     // tree uses predictor 6 or property 15
  int32_t* wp_scratch;  // 5 * 2 * (max_w + 2) ints (only if uses_wp)
  uint32_t* status = nullptr;         // device: the frame's error word
  uint64_t wp_scratch_ints = ~0ull;   // device: what the stream's slot really holds (a wider weighted-predictor channel is refused)
  uint32_t stream_id;
  uint32_t narrow_wp = 0;  // device fast path: 32-bit weighted-predictor intermediates are exact (samples of at most 12 bits)
  uint32_t slow = 0;       // the code uses prefix codes and / or LZ77: symbols are read by the general reader (tables in global memory)
  Lz77State* lz = nullptr; // LZ77 state of the stream (slow && code->lz77)
  uint32_t max_prop = 0;   // largest property index in the tree (>= 16: previous-channel properties, evaluated from `refs`)
  const ModRefs* refs = nullptr;
};

// properties 16.. (context_predict.h): |v|, v, |v - g|, v - g of the r-th reference channel at (x, y), g = clamped gradient of its W / N / NW
JXL_HD int32_t RefPropValue(const ModRefs* refs, int prop, int x, int y) {
  const int r = (prop - 16) >> 2, k = (prop - 16) & 3;
  if (!refs || r >= refs->n) return 0;
  const int32_t* rp = refs->data[r] + (size_t)y * refs->stride[r];
  const int64_t v = rp[x];
  if (k == 0) return (int32_t)(v < 0 ? -v : v);
  if (k == 1) return (int32_t)v;
  const int64_t rl = x ? rp[x - 1] : 0;
  const int64_t rt = y ? rp[x - refs->stride[r]] : rl;
  const int64_t rtl = (x && y) ? rp[x - 1 - refs->stride[r]] : rl;
  const int64_t m = rt < rl ? rt : rl, M = rt < rl ? rl : rt;
  const int64_t g = rtl < m ? M : (rtl > M ? m : rt + rl - rtl);
  const int64_t d = v - g;
  return k == 2 ? (int32_t)(d < 0 ? -d : d) : (int32_t)d;
}

// Decodes channel `chan` (index within the sub-stream, = property 0) — encoding.cc DecodeModularChannelMAANS.
// Properties >= 16 (previous-channel references) are rejected by the host before launch.
JXL_HD_NOINLINE void DecodeModularChannel(BitReader& br, AnsReader& ans, const ModularCtx& mc, const ChannelDesc& ch, int chan) {
  if (ch.w == 0 || ch.h == 0) return;
  const int w = ch.w, h = ch.h;
  const TreeNode* tree = mc.tree;
  const DevCode& code = *mc.code;
  // fast path: single-leaf tree
  WPState wps;
  if (mc.uses_wp) wps.Init(mc.wp_scratch, w);
  int32_t props[16];
  props[0] = chan; props[1] = (int32_t)mc.stream_id; props[15] = 0;
  for (int y = 0; y < h; y++) {
    int32_t* p = ch.data + (size_t)y * ch.stride;
    const int32_t* pn = p - ch.stride;
    const int32_t* pnn = pn - ch.stride;
    props[2] = y;
    props[9] = 0;
    for (int x = 0; x < w; x++) {
      const int64_t W = x ? p[x - 1] : (y ? pn[x] : 0);
      const int64_t N = y ? pn[x] : W;
      const int64_t NW = (x && y) ? pn[x - 1] : W;
      const int64_t NE = (x + 1 < w && y) ? pn[x + 1] : N;
      const int64_t WW = x > 1 ? p[x - 2] : W;
      const int64_t NN = y > 1 ? pnn[x] : N;
      const int64_t NEE = (x + 2 < w && y) ? pn[x + 2] : NE;
      props[3] = x;
      props[4] = (int32_t)Abs64(N);
      props[5] = (int32_t)Abs64(W);
      props[6] = (int32_t)N;
      props[7] = (int32_t)W;
      props[8] = (int32_t)(W - props[9]);
      props[9] = (int32_t)(W + N - NW);
      props[10] = (int32_t)(W - NW);
      props[11] = (int32_t)(NW - N);
      props[12] = (int32_t)(N - NE);
      props[13] = (int32_t)(N - NN);
      props[14] = (int32_t)(W - WW);
      int64_t wp_pred = 0;
      if (mc.uses_wp) wp_pred = wps.Predict(mc.wp, x, y, N, W, NE, NW, NN, &props[15]);
      uint32_t pos = 0;
      TreeNode n = tree[0];
      while (n.prop >= 0) {
        pos = props[n.prop & 15] > n.val ? n.a : n.b;
        n = tree[pos];
      }
      const uint32_t predictor = n.a & 0xFF, ctx = n.a >> 8;
      int64_t guess;
      switch (predictor) {
        case 0: guess = 0; break;
        case 1: guess = W; break;
        case 2: guess = N; break;
        case 3: guess = (W + N) / 2; break;
        case 4: { const int64_t pp = W + N - NW; guess = Abs64(pp - W) < Abs64(pp - N) ? W : N; } break;
        case 5: guess = ClampedGradient(N, W, NW); break;
        case 6: guess = (wp_pred + 3) >> 3; break;
        case 7: guess = NE; break;
        case 8: guess = NW; break;
        case 9: guess = WW; break;
        case 10: guess = (W + NW) / 2; break;
        case 11: guess = (N + NW) / 2; break;
        case 12: guess = (N + NE) / 2; break;
        default: guess = (6 * N - 2 * NN + 7 * W + WW + NEE + 3 * NE + 8) / 16; break;
      }
      const uint32_t tok = ReadHybridUint(br, ans, code, ctx);
      const int64_t val = (int64_t)UnpackSigned(tok) * (int64_t)n.b + (int64_t)n.val + guess;
      p[x] = (int32_t)val;
      if (mc.uses_wp) wps.Update(p[x], x, y);
    }
  }
}

// ---- VarDCT block geometry ------------------------------------------------------------------------------------------
#if defined(__HIP_DEVICE_COMPILE__)
#define JXL_TABLE __device__ static const
#else
#define JXL_TABLE static const
#endif

// Per-strategy tables packed as 4-bit fields of two 64-bit immediates (strategies 0..15 / 16..31): the lookups are
// branch-free — the switch statements these replace compiled into long chains of divergent branches on the GPU.
namespace strat {
constexpr uint8_t kLog2Cx[32] = {0, 0, 0, 0, 1, 2, 0, 1, 0, 2, 1, 2, 0, 0, 0, 0, 0, 0, 3, 2, 3, 4, 3, 4, 5, 4, 5};
constexpr uint8_t kLog2Cy[32] = {0, 0, 0, 0, 1, 2, 1, 0, 2, 0, 2, 1, 0, 0, 0, 0, 0, 0, 3, 3, 2, 4, 4, 3, 5, 5, 4};
constexpr uint8_t kOrder[32] = {0, 1, 1, 1, 2, 3, 4, 4, 5, 5, 6, 6, 1, 1, 1, 1, 1, 1, 7, 8, 8, 9, 10, 10, 11, 12, 12};
constexpr uint8_t kQuant[32] = {0, 1, 2, 3, 4, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 10, 10, 11, 12, 12, 13, 14, 14, 15, 16, 16};
constexpr uint64_t Pack(const uint8_t (&t)[32], int lo, int bits) {
  uint64_t r = 0;
  for (int i = 0; i < 64 / bits; i++) if (lo + i < 32) r |= (uint64_t)t[lo + i] << (bits * i);
  return r;
}
constexpr uint64_t kCxLo = Pack(kLog2Cx, 0, 4), kCxHi = Pack(kLog2Cx, 16, 4);
constexpr uint64_t kCyLo = Pack(kLog2Cy, 0, 4), kCyHi = Pack(kLog2Cy, 16, 4);
constexpr uint64_t kOrdLo = Pack(kOrder, 0, 4), kOrdHi = Pack(kOrder, 16, 4);
// quant kinds reach 16: 5-bit fields, 12 per word
constexpr uint64_t kQk0 = Pack(kQuant, 0, 5), kQk1 = Pack(kQuant, 12, 5), kQk2 = Pack(kQuant, 24, 5);
}  // namespace strat
JXL_HD uint32_t StratNibble(uint64_t lo, uint64_t hi, uint32_t s) { return (uint32_t)((s & 16 ? hi : lo) >> ((s & 15) * 4)) & 15u; }
JXL_HD uint32_t Log2CoveredX(uint32_t s) { return StratNibble(strat::kCxLo, strat::kCxHi, s); }
JXL_HD uint32_t Log2CoveredY(uint32_t s) { return StratNibble(strat::kCyLo, strat::kCyHi, s); }
JXL_HD uint32_t CoveredX(uint32_t s) { return 1u << Log2CoveredX(s); }
JXL_HD uint32_t CoveredY(uint32_t s) { return 1u << Log2CoveredY(s); }
JXL_HD uint32_t OrderBucket(uint32_t s) { return StratNibble(strat::kOrdLo, strat::kOrdHi, s); }
JXL_HD uint32_t QuantKind(uint32_t s) {
  const uint64_t w = s < 12 ? strat::kQk0 : s < 24 ? strat::kQk1 : strat::kQk2;
  const uint32_t i = s < 12 ? s : s < 24 ? s - 12 : s - 24;
  return (uint32_t)(w >> (i * 5)) & 31u;
}

// per-8x8-block info word written by the LF stage:
//  bits 0..4 strategy, 5 is_first, 8..15 hf_mul-1, 16..20 ix (block column inside varblock), 21..25 iy, 26..28 sharpness
JXL_HD uint32_t PackBlockInfo(uint32_t strategy, uint32_t first, uint32_t hf_mul_m1, uint32_t ix, uint32_t iy, uint32_t sharp) {
  return strategy | (first << 5) | (hf_mul_m1 << 8) | (ix << 16) | (iy << 21) | (sharp << 26);
}
JXL_HD uint32_t BI_Strategy(uint32_t v) { return v & 31; }
JXL_HD uint32_t BI_First(uint32_t v) { return (v >> 5) & 1; }
JXL_HD uint32_t BI_HfMul(uint32_t v) { return ((v >> 8) & 0xFF) + 1; }
JXL_HD uint32_t BI_Ix(uint32_t v) { return (v >> 16) & 31; }
JXL_HD uint32_t BI_Iy(uint32_t v) { return (v >> 21) & 31; }
JXL_HD uint32_t BI_Sharp(uint32_t v) { return (v >> 26) & 7; }

// block-context map as uploaded per frame (ac_context.h)
struct BlockCtxDev {
  int32_t lf_thr[3][16];   // thresholds per channel X,Y,B
  uint32_t n_lf_thr[3];
  uint32_t qf_thr[16];
  uint32_t n_qf_thr;
  uint32_t num_lf_ctxs, num_ctxs;
  uint8_t ctx_map[3 * 13 * 64];
};

// ---- float samples of Modular images (dec_modular.cc int_to_float): the integer holds the bits of a float with `exp_bits` exponent
// bits out of `bits`; repacked into binary32 (subnormals of the narrow format are normalised, bits == 32 is a plain reinterpretation)
JXL_HD float IntToFloatSample(int32_t v, uint32_t bits, uint32_t exp_bits) {
  uint32_t f = (uint32_t)v;
  if (bits == 32) { float r; memcpy(&r, &f, 4); return r; }
  const int exp_bias = (1 << (exp_bits - 1)) - 1;
  const uint32_t sign_shift = bits - 1, mant_bits = bits - exp_bits - 1, mant_shift = 23 - mant_bits;
  const uint32_t signbit = (f >> sign_shift) & 1u;
  f &= (1u << sign_shift) - 1u;
  if (f == 0) return signbit ? -0.0f : 0.0f;
  int exp = (int)(f >> mant_bits);
  uint32_t mantissa = (f & ((1u << mant_bits) - 1u)) << mant_shift;
  if (exp == 0 && exp_bits < 8) {                 // subnormal of the narrow format: normalise, the leading 1 becomes implicit
    while ((mantissa & 0x800000u) == 0) { mantissa <<= 1; exp--; }
    exp++;
    mantissa &= 0x7FFFFFu;
  }
  exp = exp - exp_bias + 127;
  const uint32_t out = (signbit ? 0x80000000u : 0u) | ((uint32_t)exp << 23) | mantissa;
  float r; memcpy(&r, &out, 4);
  return r;
}

// ---- HDR transfer functions of the output stage (stage_from_linear.cc OpPq / OpHlg) ----------------------------------------------------
// TF_PQ::EncodedFromDisplay: 4-over-4 rational polynomials in x^(1/4) (a second pair below 1e-4), x = linear value x intensity_target / 10000
JXL_HD float PqFromLinear(float v, float scale) {
  const float xs = fabsf(v) * scale;
  const float t = sqrtf(sqrtf(xs));
  float yp, yq;
  if (xs < 1e-4f) {
    yp = fmaf(fmaf(fmaf(fmaf(-2.864824e+05f, t, 6.889862e+04f), t, 1.352821e+02f), t, 3.881234e-01f), t, 9.863406e-06f);
    yq = fmaf(fmaf(fmaf(fmaf(-2.072546e+05f, t, -4.389884e+04f), t, 1.608477e+04f), t, 1.477719e+03f), t, 3.371868e+01f);
  } else {
    yp = fmaf(fmaf(fmaf(fmaf(4.838434e+01f, t, 1.492516e+02f), t, 5.522776e+01f), t, -1.095778e+00f), t, 1.351392e-02f);
    yq = fmaf(fmaf(fmaf(fmaf(2.590418e+01f, t, 1.120607e+02f), t, 9.263710e+01f), t, 2.016708e+01f), t, 1.012416e+00f);
  }
  return copysignf(yp / yq, v);
}
// TF_HLG_Base::EncodedFromDisplay (scalar double in libjxl: OpHlg goes lane by lane)
JXL_HD float HlgFromLinear(float v) {
  const double kA = 0.17883277, kB = 1 - 4 * kA, kC = 0.5599107295, kDiv12 = 1.0 / 12;
  const double s = fabs((double)v);
  if (s == 0.0) return 0.0f;
  const double e = s <= kDiv12 ? sqrt(3.0 * s) : kA * log(12 * s - kB) + kC;
  return (float)copysign(e, (double)v);
}
// HlgOOTF::Apply (cms/tone_mapping-inl.h): display light -> scene light; par = {exponent, apply?, luminances of the output primaries}
template <typename PowFn> JXL_HD void HlgInverseOotf(const float* par, float& r, float& g, float& b, PowFn fast_powf) {
  if (par[1] == 0.0f) return;
  const float luminance = fmaf(par[2], r, fmaf(par[3], g, par[4] * b));
  const float ratio = fminf(fast_powf(luminance, par[0]), 1e9f);
  r *= ratio; g *= ratio; b *= ratio;
}

}  // namespace jxlhip
