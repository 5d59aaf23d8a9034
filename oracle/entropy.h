// ORACLE — TEST INFRASTRUCTURE ONLY (see bits.h).
// Entropy decoder of JPEG XL: restates libjxl v0.11.2 lib/jxl/dec_ans.{h,cc}, dec_huffman.cc,
// dec_context_map.cc, ans_common.cc (alias table).  SURVEY.md App. B.4 — [V] on all reference fixtures
// except the LZ77 branch, which no fixture uses ([R]).
#pragma once
#include "bits.h"
#include <algorithm>

namespace jxlo {

struct HybridUintConfig {
  uint32_t split_exponent = 4, msb_in_token = 2, lsb_in_token = 0, split_token = 16;
};

// dec_ans.cc DecodeUintConfig
inline HybridUintConfig ReadUintConfig(BitReader& br, int log_alpha) {
  HybridUintConfig c;
  c.split_exponent = br.u(CeilLog2(log_alpha + 1));
  c.msb_in_token = 0; c.lsb_in_token = 0;
  if (c.split_exponent != (uint32_t)log_alpha) {
    c.msb_in_token = br.u(CeilLog2(c.split_exponent + 1));
    if (c.msb_in_token > c.split_exponent) JXLO_FAIL("bad msb_in_token");
    c.lsb_in_token = br.u(CeilLog2(c.split_exponent - c.msb_in_token + 1));
  }
  if (c.lsb_in_token + c.msb_in_token > c.split_exponent) JXLO_FAIL("bad lsb_in_token");
  c.split_token = 1u << c.split_exponent;
  return c;
}

// ans_common.cc InitAliasTable — exact construction order matters (SURVEY B.4 [V]).
struct AliasEntry { uint8_t cutoff, right; uint16_t freq0, offs1, freq1; };

struct PrefixCode {
  // canonical LSB-first code: (len,symbol) lookup by up-to-15-bit peek
  std::vector<uint16_t> lut_sym;  // size 1<<maxlen
  std::vector<uint8_t> lut_len;
  int maxlen = 0;
  int single = -1;  // >=0: zero-bit single symbol
};

inline uint32_t VarLenUint8(BitReader& br) {
  if (!br.u(1)) return 0;
  int n = br.u(3);
  if (n == 0) return 1;
  return br.u(n) + (1u << n);
}
inline uint32_t VarLenUint16(BitReader& br) {
  if (!br.u(1)) return 0;
  int n = br.u(4);
  if (n == 0) return 1;
  return br.u(n) + (1u << n);
}

// dec_ans.cc ReadHistogram — distribution summing to 4096 (ANS_LOG_TAB_SIZE = 12)
inline std::vector<int> ReadANSHistogram(BitReader& br) {
  std::vector<int> counts;
  if (br.u(1)) {  // simple code
    int ns = br.u(1) + 1;
    uint32_t s0 = VarLenUint8(br);
    if (ns == 1) {
      counts.assign(s0 + 1, 0);
      counts[s0] = 4096;
    } else {
      uint32_t s1 = VarLenUint8(br);
      if (s0 == s1) JXLO_FAIL("simple ANS histogram with equal symbols");
      counts.assign(std::max(s0, s1) + 1, 0);
      counts[s0] = br.u(12);
      counts[s1] = 4096 - counts[s0];
    }
    return counts;
  }
  if (br.u(1)) {  // flat
    int n = VarLenUint8(br) + 1;
    counts.assign(n, 4096 / n);
    for (int i = 0; i < 4096 % n; i++) counts[i]++;
    return counts;
  }
  int len = 0;
  while (len < 3 && br.u(1)) len++;
  int shift = (int)(br.u(len) | (1u << len)) - 1;
  if (shift > 13) JXLO_FAIL("bad ANS shift");
  int length = VarLenUint8(br) + 3;
  // log-count prefix code, 7-bit peek LUT (dec_ans.cc kLogCountLut)
  static const uint8_t lut[128][2] = {
      {3, 10}, {7, 12}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {6, 11}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {7, 13}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {6, 11}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
      {3, 10}, {5, 0},  {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2},
  };
  std::vector<int> logcounts(length, 0), same(length, 0);
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < length; i++) {
    uint32_t idx = (uint32_t)br.peek(7);
    br.skip(lut[idx][0]);
    if (br.pos > br.size * 8) JXLO_FAIL("overrun");
    logcounts[i] = lut[idx][1];
    if (logcounts[i] == 13) {  // RLE
      int rl = VarLenUint8(br);
      same[i] = rl + 5;
      i += rl + 3;
      continue;
    }
    if (logcounts[i] > omit_log) { omit_log = logcounts[i]; omit_pos = i; }
  }
  if (omit_pos < 0) JXLO_FAIL("ANS histogram without omit position");
  if (omit_pos + 1 < length && logcounts[omit_pos + 1] == 13) JXLO_FAIL("RLE after omit pos");
  counts.assign(length, 0);
  int total = 0, prev = 0, numsame = 0;
  for (int i = 0; i < length; i++) {
    if (same[i]) { numsame = same[i] - 1; prev = i > 0 ? counts[i - 1] : 0; }
    if (numsame > 0) {
      counts[i] = prev;
      numsame--;
    } else {
      int code = logcounts[i];
      if (i == omit_pos) continue;
      if (code == 0) continue;
      if (code == 1) counts[i] = 1;
      else {
        int bitcount = std::min(std::max(0, shift - ((12 - code + 1) >> 1)), code - 1);
        counts[i] = (1 << (code - 1)) + (br.u(bitcount) << (code - 1 - bitcount));
      }
    }
    total += counts[i];
  }
  counts[omit_pos] = 4096 - total;
  if (counts[omit_pos] <= 0) JXLO_FAIL("ANS histogram omit count <= 0");
  return counts;
}

inline void BuildAliasTable(std::vector<int> dist, int log_alpha, std::vector<AliasEntry>& out) {
  const int T = 1 << log_alpha, B = 4096 >> log_alpha;
  while (!dist.empty() && dist.back() == 0) dist.pop_back();
  if (dist.empty()) { dist.assign(1, 4096); }
  if ((int)dist.size() > T) JXLO_FAIL("alphabet larger than table");
  out.assign(T, AliasEntry{});
  for (size_t s = 0; s < dist.size(); s++) {
    if (dist[s] == 4096) {
      for (int i = 0; i < T; i++) {
        out[i].cutoff = 0; out[i].right = (uint8_t)s; out[i].freq0 = 0; out[i].offs1 = (uint16_t)(B * i); out[i].freq1 = 4096;
      }
      return;
    }
  }
  std::vector<int> cut(T, 0), right(T, 0), offs1(T, 0);
  for (size_t i = 0; i < dist.size(); i++) cut[i] = dist[i];
  std::vector<int> over, under;
  for (int i = 0; i < T; i++) {
    if (cut[i] > B) over.push_back(i);
    else if (cut[i] < B) under.push_back(i);
  }
  while (!over.empty()) {
    if (under.empty()) JXLO_FAIL("alias table construction failed");
    int o = over.back(); over.pop_back();
    int u = under.back(); under.pop_back();
    int by = B - cut[u];
    cut[o] -= by;
    right[u] = o;
    offs1[u] = cut[o];
    if (cut[o] < B) under.push_back(o);
    else if (cut[o] > B) over.push_back(o);
  }
  for (int i = 0; i < T; i++) {
    if (cut[i] == B) { right[i] = i; offs1[i] = 0; cut[i] = 0; }
    else offs1[i] -= cut[i];
    out[i].cutoff = (uint8_t)cut[i];
    out[i].right = (uint8_t)right[i];
    out[i].freq0 = (uint16_t)(i < (int)dist.size() ? dist[i] : 0);
    out[i].offs1 = (uint16_t)offs1[i];
    out[i].freq1 = (uint16_t)(right[i] < (int)dist.size() ? dist[right[i]] : 0);
  }
}

// dec_huffman.cc (Brotli RFC 7932 §3.4/3.5 prefix codes, LSB-first)
inline void BuildPrefixLut(const std::vector<uint8_t>& lens, PrefixCode& pc) {
  int maxlen = 0, nonzero = 0, last = -1;
  for (size_t i = 0; i < lens.size(); i++) if (lens[i]) { maxlen = std::max<int>(maxlen, lens[i]); nonzero++; last = (int)i; }
  if (nonzero == 0) { pc.single = 0; pc.maxlen = 0; return; }
  if (nonzero == 1) { pc.single = last; pc.maxlen = 0; return; }
  pc.maxlen = maxlen;
  pc.lut_sym.assign(1u << maxlen, 0);
  pc.lut_len.assign(1u << maxlen, 0);
  uint32_t code = 0;
  for (int len = 1; len <= maxlen; len++) {
    for (size_t s = 0; s < lens.size(); s++) {
      if (lens[s] != len) continue;
      // bit-reverse code of `len` bits
      uint32_t rev = 0;
      for (int b = 0; b < len; b++) if (code >> b & 1) rev |= 1u << (len - 1 - b);
      for (uint32_t k = rev; k < (1u << maxlen); k += 1u << len) { pc.lut_sym[k] = (uint16_t)s; pc.lut_len[k] = (uint8_t)len; }
      code++;
    }
    code <<= 1;
  }
}

inline void ReadPrefixCode(BitReader& br, int alphabet_size, PrefixCode& pc) {
  if (alphabet_size == 1) { pc.single = 0; return; }
  std::vector<uint8_t> lens(alphabet_size, 0);
  int hskip = br.u(2);
  if (hskip == 1) {  // simple
    int max_bits = 0;
    { int v = alphabet_size - 1; while (v) { max_bits++; v >>= 1; } }
    int nsym = br.u(2) + 1;
    int syms[4];
    for (int i = 0; i < nsym; i++) { syms[i] = br.u(max_bits); if (syms[i] >= alphabet_size) JXLO_FAIL("bad simple prefix symbol"); }
    for (int i = 0; i < nsym; i++) for (int j = i + 1; j < nsym; j++) if (syms[i] == syms[j]) JXLO_FAIL("duplicate simple prefix symbol");
    if (nsym == 1) { pc.single = syms[0]; return; }
    if (nsym == 2) { lens[syms[0]] = 1; lens[syms[1]] = 1; }
    else if (nsym == 3) { lens[syms[0]] = 1; lens[syms[1]] = 2; lens[syms[2]] = 2; }
    else {
      if (br.u(1)) { lens[syms[0]] = 1; lens[syms[1]] = 2; lens[syms[2]] = 3; lens[syms[3]] = 3; }
      else { for (int i = 0; i < 4; i++) lens[syms[i]] = 2; }
    }
    BuildPrefixLut(lens, pc);
    return;
  }
  static const uint8_t kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kLen[16] = {2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4};
  static const uint8_t kVal[16] = {0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5};
  std::vector<uint8_t> cl(18, 0);
  int space = 32, num_codes = 0;
  for (int i = hskip; i < 18 && space > 0; i++) {
    uint32_t p = (uint32_t)br.peek(4);
    br.skip(kLen[p]);
    int v = kVal[p];
    cl[kOrder[i]] = (uint8_t)v;
    if (v) { space -= 32 >> v; num_codes++; }
  }
  if (num_codes != 1 && space != 0) JXLO_FAIL("bad code length code");
  PrefixCode clc;
  BuildPrefixLut(cl, clc);
  int symbol = 0, prev_len = 8, repeat = 0, repeat_len = 0;
  int sp = 32768;
  while (symbol < alphabet_size && sp > 0) {
    int v;
    if (clc.single >= 0) v = clc.single;
    else { uint32_t p = (uint32_t)br.peek(clc.maxlen); v = clc.lut_sym[p]; br.skip(clc.lut_len[p]); }
    if (v < 16) {
      repeat = 0;
      lens[symbol++] = (uint8_t)v;
      if (v) { prev_len = v; sp -= 32768 >> v; }
    } else {
      int extra = v == 16 ? 2 : 3;
      int new_len = v == 16 ? prev_len : 0;
      if (repeat_len != new_len) { repeat = 0; repeat_len = new_len; }
      int old = repeat;
      if (repeat > 0) { repeat -= 2; repeat <<= extra; }
      repeat += br.u(extra) + 3;
      int delta = repeat - old;
      if (symbol + delta > alphabet_size) JXLO_FAIL("prefix code repeat overflow");
      for (int i = 0; i < delta; i++) lens[symbol++] = (uint8_t)repeat_len;
      if (repeat_len) sp -= delta << (15 - repeat_len);
    }
  }
  if (sp != 0) JXLO_FAIL("prefix code space != 0");
  BuildPrefixLut(lens, pc);
}

struct EntropyCode {
  bool lz77 = false;
  uint32_t lz_min_symbol = 0, lz_min_length = 0;
  HybridUintConfig lz_len_cfg;
  std::vector<uint8_t> ctx_map;  // context -> cluster (includes the LZ77 distance context as last)
  int num_clusters = 0;
  bool use_prefix = false;
  int log_alpha = 0;
  std::vector<HybridUintConfig> cfg;
  std::vector<std::vector<AliasEntry>> alias;  // ANS
  std::vector<PrefixCode> prefix;
  std::vector<std::vector<int>> dists;  // kept for tests
};

struct EntropyCode;
inline void ReadEntropyCode(BitReader& br, int num_ctx, EntropyCode& ec, bool allow_lz77 = true);

// stateful reader of one stream (dec_ans.h ANSSymbolReader)
struct SymbolReader {
  const EntropyCode* ec = nullptr;
  uint32_t state = 0;
  bool ans_init = false;
  // LZ77
  std::vector<uint32_t> window;
  uint32_t num_to_copy = 0, copy_pos = 0, num_decoded = 0;
  uint32_t dist_multiplier = 0;
  size_t tokens = 0;
  static constexpr uint32_t kWindow = 1u << 20, kMask = kWindow - 1;

  void Init(const EntropyCode* e, BitReader& br, uint32_t dist_mult = 0) {
    ec = e; dist_multiplier = dist_mult;
    if (!e->use_prefix) { state = br.u(32); ans_init = true; }
    if (e->lz77) window.assign(kWindow, 0);
    num_to_copy = copy_pos = num_decoded = 0;
  }
  inline uint32_t ReadSymbolCluster(BitReader& br, int cluster) {
    tokens++;
    if (ec->use_prefix) {
      const PrefixCode& pc = ec->prefix[cluster];
      if (pc.single >= 0) return pc.single;
      uint32_t p = (uint32_t)br.peek(pc.maxlen);
      br.skip(pc.lut_len[p]);
      if (br.pos > br.size * 8) JXLO_FAIL("overrun");
      return pc.lut_sym[p];
    }
    const int la = ec->log_alpha;
    const uint32_t res = state & 0xFFF;
    const uint32_t i = res >> (12 - la), pos = res & ((1u << (12 - la)) - 1);
    const AliasEntry& e = ec->alias[cluster][i];
    bool hit = pos >= e.cutoff;
    uint32_t sym = hit ? e.right : i;
    uint32_t off = hit ? e.offs1 + pos : pos;
    uint32_t freq = hit ? e.freq1 : e.freq0;
    state = freq * (state >> 12) + off;
    if (state < (1u << 16)) state = (state << 16) | br.u(16);
    return sym;
  }
  static inline uint32_t ReadHybrid(const HybridUintConfig& c, uint32_t tok, BitReader& br) {
    if (tok < c.split_token) return tok;
    uint32_t nbits = c.split_exponent - (c.msb_in_token + c.lsb_in_token) + ((tok - c.split_token) >> (c.msb_in_token + c.lsb_in_token));
    if (nbits > 32) JXLO_FAIL("hybrid uint too many bits");
    uint32_t low = tok & ((1u << c.lsb_in_token) - 1);
    tok >>= c.lsb_in_token;
    uint32_t bits = nbits ? (uint32_t)(nbits <= 24 ? br.u(nbits) : (br.u(16) | ((uint32_t)br.u(nbits - 16) << 16))) : 0;
    uint64_t hi = (uint64_t)((1u << c.msb_in_token) | (tok & ((1u << c.msb_in_token) - 1)));
    uint64_t ret = (((hi << nbits) | bits) << c.lsb_in_token) | low;
    return (uint32_t)ret;
  }
  uint32_t Read(BitReader& br, int ctx);
  bool CheckFinal() const { return ec->use_prefix || state == 0x130000u; }
};

static const int8_t kSpecialDistances[120][2] = {
    {0, 1},  {1, 0},  {1, 1},  {-1, 1}, {0, 2},  {2, 0},  {1, 2},  {-1, 2}, {2, 1},  {-2, 1}, {2, 2},  {-2, 2}, {0, 3},  {3, 0},  {1, 3},
    {-1, 3}, {3, 1},  {-3, 1}, {2, 3},  {-2, 3}, {3, 2},  {-3, 2}, {0, 4},  {4, 0},  {1, 4},  {-1, 4}, {4, 1},  {-4, 1}, {3, 3},  {-3, 3},
    {2, 4},  {-2, 4}, {4, 2},  {-4, 2}, {0, 5},  {3, 4},  {-3, 4}, {4, 3},  {-4, 3}, {5, 0},  {1, 5},  {-1, 5}, {5, 1},  {-5, 1}, {2, 5},
    {-2, 5}, {5, 2},  {-5, 2}, {4, 4},  {-4, 4}, {3, 5},  {-3, 5}, {5, 3},  {-5, 3}, {0, 6},  {6, 0},  {1, 6},  {-1, 6}, {6, 1},  {-6, 1},
    {2, 6},  {-2, 6}, {6, 2},  {-6, 2}, {4, 5},  {-4, 5}, {5, 4},  {-5, 4}, {3, 6},  {-3, 6}, {6, 3},  {-6, 3}, {0, 7},  {7, 0},  {1, 7},
    {-1, 7}, {5, 5},  {-5, 5}, {7, 1},  {-7, 1}, {4, 6},  {-4, 6}, {6, 4},  {-6, 4}, {2, 7},  {-2, 7}, {7, 2},  {-7, 2}, {3, 7},  {-3, 7},
    {7, 3},  {-7, 3}, {5, 6},  {-5, 6}, {6, 5},  {-6, 5}, {8, 0},  {4, 7},  {-4, 7}, {7, 4},  {-7, 4}, {8, 1},  {8, 2},  {6, 6},  {-6, 6},
    {8, 3},  {5, 7},  {-5, 7}, {7, 5},  {-7, 5}, {8, 4},  {6, 7},  {-6, 7}, {7, 6},  {-7, 6}, {8, 5},  {7, 7},  {-7, 7}, {8, 6},  {8, 7}};

inline uint32_t SymbolReader::Read(BitReader& br, int ctx) {
  if (!ec->lz77) {
    int cl = ec->ctx_map[ctx];
    uint32_t tok = ReadSymbolCluster(br, cl);
    return ReadHybrid(ec->cfg[cl], tok, br);
  }
  // LZ77 [R] (dec_ans.h ReadHybridUintClustered with lz77 enabled)
  for (;;) {
    if (num_to_copy > 0) {
      uint32_t r = window[(copy_pos++) & kMask];
      num_to_copy--;
      window[(num_decoded++) & kMask] = r;
      return r;
    }
    int cl = ec->ctx_map[ctx];
    uint32_t tok = ReadSymbolCluster(br, cl);
    if (tok >= ec->lz_min_symbol) {
      num_to_copy = ReadHybrid(ec->lz_len_cfg, tok - ec->lz_min_symbol, br) + ec->lz_min_length;
      int dcl = ec->ctx_map.back();
      uint32_t dtok = ReadSymbolCluster(br, dcl);
      uint32_t distance = ReadHybrid(ec->cfg[dcl], dtok, br);
      uint32_t nspecial = dist_multiplier == 0 ? 0 : 120;
      if (distance < nspecial) {
        int d = kSpecialDistances[distance][0] + (int)dist_multiplier * kSpecialDistances[distance][1];
        distance = d < 1 ? 1 : (uint32_t)d;
      } else {
        distance = distance + 1 - nspecial;
      }
      if (distance > num_decoded) distance = num_decoded;
      if (distance > kWindow) distance = kWindow;
      copy_pos = num_decoded - distance;
      if (distance == 0) {
        // no history: libjxl fills with zeros
        uint32_t n = std::min<uint32_t>(num_to_copy, kWindow);
        for (uint32_t i = 0; i < n; i++) window[i] = 0;
      }
      if (num_to_copy < ec->lz_min_length) JXLO_FAIL("lz77 length overflow");
      continue;
    }
    uint32_t r = ReadHybrid(ec->cfg[cl], tok, br);
    window[(num_decoded++) & kMask] = r;
    return r;
  }
}

// dec_context_map.cc DecodeContextMap
inline void ReadContextMap(BitReader& br, int num_ctx, std::vector<uint8_t>& map, int& num_clusters) {
  map.assign(num_ctx, 0);
  bool simple = br.Bool();
  if (simple) {
    int bits = br.u(2);
    for (int i = 0; i < num_ctx; i++) map[i] = (uint8_t)br.u(bits);
  } else {
    bool use_mtf = br.Bool();
    EntropyCode nested;
    ReadEntropyCode(br, 1, nested, /*allow_lz77=*/num_ctx > 2);
    SymbolReader sr;
    sr.Init(&nested, br);
    for (int i = 0; i < num_ctx; i++) {
      uint32_t v = sr.Read(br, 0);
      if (v > 255) JXLO_FAIL("context map value too large");
      map[i] = (uint8_t)v;
    }
    if (!sr.CheckFinal()) JXLO_FAIL("context map ANS final state");
    if (use_mtf) {
      uint8_t mtf[256];
      for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
      for (int i = 0; i < num_ctx; i++) {
        uint8_t idx = map[i];
        uint8_t v = mtf[idx];
        map[i] = v;
        for (int j = idx; j > 0; j--) mtf[j] = mtf[j - 1];
        mtf[0] = v;
      }
    }
  }
  int mx = 0;
  for (int i = 0; i < num_ctx; i++) mx = std::max<int>(mx, map[i]);
  num_clusters = mx + 1;
  // every cluster id in [0,max] must be used
  std::vector<bool> used(num_clusters, false);
  for (int i = 0; i < num_ctx; i++) used[map[i]] = true;
  for (int i = 0; i < num_clusters; i++) if (!used[i]) JXLO_FAIL("context map skips a cluster");
}

// dec_ans.cc DecodeHistograms
inline void ReadEntropyCode(BitReader& br, int num_ctx, EntropyCode& ec, bool allow_lz77) {
  ec = EntropyCode();
  ec.lz77 = br.Bool();
  if (ec.lz77) {
    if (!allow_lz77) JXLO_FAIL("lz77 not allowed here");
    ec.lz_min_symbol = U32(br, Val(224), Val(512), Val(4096), BitsOffset(15, 8));
    ec.lz_min_length = U32(br, Val(3), Val(4), BitsOffset(2, 5), BitsOffset(8, 9));
    ec.lz_len_cfg = ReadUintConfig(br, 8);
    num_ctx += 1;
  }
  if (num_ctx > 1) ReadContextMap(br, num_ctx, ec.ctx_map, ec.num_clusters);
  else { ec.ctx_map.assign(1, 0); ec.num_clusters = 1; }
  ec.use_prefix = br.Bool();
  ec.log_alpha = ec.use_prefix ? 15 : 5 + br.u(2);
  ec.cfg.resize(ec.num_clusters);
  for (int i = 0; i < ec.num_clusters; i++) ec.cfg[i] = ReadUintConfig(br, ec.log_alpha);
  if (ec.use_prefix) {
    std::vector<int> asz(ec.num_clusters);
    for (int i = 0; i < ec.num_clusters; i++) asz[i] = VarLenUint16(br) + 1;
    ec.prefix.resize(ec.num_clusters);
    for (int i = 0; i < ec.num_clusters; i++) {
      if (asz[i] > (1 << 15)) JXLO_FAIL("prefix alphabet too large");
      ReadPrefixCode(br, asz[i], ec.prefix[i]);
    }
  } else {
    ec.alias.resize(ec.num_clusters);
    ec.dists.resize(ec.num_clusters);
    for (int i = 0; i < ec.num_clusters; i++) {
      ec.dists[i] = ReadANSHistogram(br);
      BuildAliasTable(ec.dists[i], ec.log_alpha, ec.alias[i]);
    }
  }
}

}  // namespace jxlo
