// jxlsynth — bit-stream synthesiser (fixture/bench input generator; NOT part of the product decode path and
// independent of oracle/).  Entropy *encoder* side of JPEG XL: bit writer, hybrid-uint tokens, ANS histograms in the
// codestream format, alias-table-consistent rANS encoding, context clustering and context-map coding.
// The format facts mirror SURVEY.md App. B.4 (verified against the reference's fixtures by the decoder side).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace synth {

struct BitWriter {
  std::vector<uint8_t> bytes;
  uint64_t acc = 0;
  int nacc = 0;
  void put(uint64_t v, int n) {  // n <= 32
    if (n == 0) return;
    acc |= (v & ((n >= 64) ? ~0ull : ((1ull << n) - 1))) << nacc;
    nacc += n;
    while (nacc >= 8) { bytes.push_back((uint8_t)acc); acc >>= 8; nacc -= 8; }
  }
  void align() { if (nacc) { bytes.push_back((uint8_t)acc); acc = 0; nacc = 0; } }
  size_t bits() const { return bytes.size() * 8 + nacc; }
  void append(const BitWriter& o) {  // o may be unaligned
    for (uint8_t b : o.bytes) put(b, 8);
    if (o.nacc) put(o.acc, o.nacc);
  }
};

inline int CeilLog2(uint32_t x) { int r = 0; while ((1ull << r) < x) r++; return r; }
inline int FloorLog2(uint32_t x) { int r = 0; while (x >>= 1) r++; return r; }
inline uint32_t PackSigned(int32_t v) { return v >= 0 ? (uint32_t)v * 2 : (uint32_t)(-(int64_t)v) * 2 - 1; }

// U32 field writer: picks the first distribution that can represent v
struct Dist { int bits; uint32_t off; };
inline void WriteU32(BitWriter& w, uint32_t v, Dist d0, Dist d1, Dist d2, Dist d3) {
  Dist d[4] = {d0, d1, d2, d3};
  for (int i = 0; i < 4; i++) {
    if (v >= d[i].off && (uint64_t)(v - d[i].off) < (1ull << d[i].bits)) { w.put(i, 2); w.put(v - d[i].off, d[i].bits); return; }
  }
  throw std::runtime_error("U32 value not representable");
}
inline void WriteU64(BitWriter& w, uint64_t v) {
  if (v == 0) { w.put(0, 2); return; }
  if (v <= 16) { w.put(1, 2); w.put(v - 1, 4); return; }
  if (v <= 272) { w.put(2, 2); w.put(v - 17, 8); return; }
  w.put(3, 2);
  w.put(v & 0xFFF, 12);
  v >>= 12;
  int shift = 12;
  while (v) {
    w.put(1, 1);
    if (shift == 60) { w.put(v & 0xF, 4); return; }
    w.put(v & 0xFF, 8);
    v >>= 8; shift += 8;
  }
  w.put(0, 1);
}
inline uint16_t FloatToHalfBits(float f) {
  uint32_t x; memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000;
  int32_t exp = (int32_t)((x >> 23) & 0xFF) - 127 + 15;
  uint32_t mant = x & 0x7FFFFF;
  if (exp >= 31) throw std::runtime_error("F16 overflow");
  if (exp <= 0) {
    if (exp < -10) return (uint16_t)sign;
    mant |= 0x800000;
    int shift = 14 - exp;
    uint32_t m = mant >> shift, rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (m & 1))) m++;
    return (uint16_t)(sign | m);
  }
  uint32_t m = mant >> 13, rem = mant & 0x1FFF;
  uint32_t r = (uint32_t)(exp << 10) | m;
  if (rem > 0x1000 || (rem == 0x1000 && (m & 1))) r++;
  if ((r >> 10) >= 31) throw std::runtime_error("F16 overflow");
  return (uint16_t)(sign | r);
}
inline float HalfBitsToFloat(uint16_t b) {
  uint32_t sign = b >> 15, exp = (b >> 10) & 31, mant = b & 1023;
  float v = exp == 0 ? std::ldexp((float)mant, -24) : std::ldexp((float)(mant + 1024), (int)exp - 25);
  return sign ? -v : v;
}
inline float RoundToHalf(float f) { return HalfBitsToFloat(FloatToHalfBits(f)); }
inline void WriteF16(BitWriter& w, float f) { w.put(FloatToHalfBits(f), 16); }

// ---- tokens ---------------------------------------------------------------------------------------------------------
// raw = 1: `value` is the entropy-coded symbol itself, followed by `nb` literal bits `bits` (LZ77 length symbols)
struct Token { uint32_t ctx; uint32_t value; uint8_t raw = 0; uint8_t nb = 0; uint32_t bits = 0; };

struct UintConfig { int split_exponent, msb, lsb; };

inline void EncodeHybrid(const UintConfig& c, uint32_t v, uint32_t* tok, uint32_t* nbits, uint32_t* bits) {
  uint32_t split = 1u << c.split_exponent;
  if (v < split) { *tok = v; *nbits = 0; *bits = 0; return; }
  uint32_t n = FloorLog2(v);
  uint32_t m = v - (1u << n);
  *tok = split + ((n - c.split_exponent) << (c.msb + c.lsb)) + ((m >> (n - c.msb)) << c.lsb) + (m & ((1u << c.lsb) - 1));
  *nbits = n - c.msb - c.lsb;
  *bits = (m >> c.lsb) & ((1u << *nbits) - 1);
}

inline void TokenSymbol(const UintConfig& c, const Token& t, uint32_t* tok, uint32_t* nbits, uint32_t* bits) {
  if (t.raw) { *tok = t.value; *nbits = t.nb; *bits = t.bits; return; }
  EncodeHybrid(c, t.value, tok, nbits, bits);
}

// ---- ANS ---------------------------------------------------------------------------------------------------------
struct AliasE { int cutoff, right, offs1; };

// identical construction to the decoder (SURVEY B.4 [V])
inline void BuildAlias(const std::vector<int>& dist_in, int log_alpha, std::vector<AliasE>& out) {
  const int T = 1 << log_alpha, B = 4096 >> log_alpha;
  std::vector<int> dist = dist_in;
  while (!dist.empty() && dist.back() == 0) dist.pop_back();
  out.assign(T, AliasE{0, 0, 0});
  for (size_t s = 0; s < dist.size(); s++)
    if (dist[s] == 4096) { for (int i = 0; i < T; i++) out[i] = AliasE{0, (int)s, B * i}; return; }
  std::vector<int> cut(T, 0), right(T, 0), offs1(T, 0), over, under;
  for (size_t i = 0; i < dist.size(); i++) cut[i] = dist[i];
  for (int i = 0; i < T; i++) { if (cut[i] > B) over.push_back(i); else if (cut[i] < B) under.push_back(i); }
  while (!over.empty()) {
    int o = over.back(); over.pop_back();
    int u = under.back(); under.pop_back();
    cut[o] -= B - cut[u];
    right[u] = o; offs1[u] = cut[o];
    if (cut[o] < B) under.push_back(o); else if (cut[o] > B) over.push_back(o);
  }
  for (int i = 0; i < T; i++) {
    if (cut[i] == B) { right[i] = i; offs1[i] = 0; cut[i] = 0; } else offs1[i] -= cut[i];
    out[i] = AliasE{cut[i], right[i], offs1[i]};
  }
}

struct Histogram {
  std::vector<int> counts;  // normalised to 4096 after Normalize()
  std::vector<std::vector<uint16_t>> reverse;  // [symbol][offset] -> slot value in [0,4096)
};

// Normalise raw counts to sum 4096 with every used symbol >= 1 (alphabet trimmed to last used symbol).
inline void Normalize(const std::vector<uint32_t>& raw, std::vector<int>& out) {
  size_t n = raw.size();
  while (n > 0 && raw[n - 1] == 0) n--;
  out.assign(std::max<size_t>(n, 1), 0);
  uint64_t total = 0;
  for (size_t i = 0; i < n; i++) total += raw[i];
  if (total == 0) { out.assign(1, 4096); return; }
  int used = 0;
  for (size_t i = 0; i < n; i++) used += raw[i] != 0;
  if (used == 1) { for (size_t i = 0; i < n; i++) if (raw[i]) out[i] = 4096; return; }
  int sum = 0;
  size_t maxi = 0;
  for (size_t i = 0; i < n; i++) {
    if (!raw[i]) continue;
    int c = (int)((double)raw[i] * 4096.0 / (double)total + 0.5);
    if (c < 1) c = 1;
    out[i] = c; sum += c;
    if (out[i] > out[maxi] || !out[maxi]) maxi = i;
  }
  // fix the sum on the largest entries
  int diff = 4096 - sum;
  while (diff != 0) {
    size_t best = 0; int bv = -1;
    for (size_t i = 0; i < n; i++) if (out[i] > bv) { bv = out[i]; best = i; }
    if (diff > 0) { out[best] += diff; diff = 0; }
    else {
      int take = std::min(-diff, out[best] - 1);
      if (take <= 0) throw std::runtime_error("cannot normalise histogram");
      out[best] -= take; diff += take;
    }
  }
}

// dec_ans.cc kLogCountLut inverted: code for each log-count value (LSB-first)
struct LogCountCode { int nbits[14]; int bits[14]; };
inline const LogCountCode& LogCountCodes() {
  static LogCountCode c;
  static bool init = false;
  if (!init) {
    static const uint8_t base[16][2] = {{3, 10}, {7, 12}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2}};
    static const uint8_t idx1[8][2] = {{7, 12}, {5, 0}, {6, 11}, {5, 0}, {7, 13}, {5, 0}, {6, 11}, {5, 0}};
    for (int v = 0; v < 14; v++) c.nbits[v] = 0;
    for (int i = 0; i < 128; i++) {
      int nb, val;
      if ((i & 15) == 1) { nb = idx1[i >> 4][0]; val = idx1[i >> 4][1]; } else { nb = base[i & 15][0]; val = base[i & 15][1]; }
      if (c.nbits[val] == 0 && i < (1 << nb)) { c.nbits[val] = nb; c.bits[val] = i; }
    }
    init = true;
  }
  return c;
}

inline void WriteVarLenUint8(BitWriter& w, uint32_t n) {
  if (n == 0) { w.put(0, 1); return; }
  w.put(1, 1);
  int nb = FloorLog2(n);
  w.put(nb, 3);
  w.put(n - (1u << nb), nb);
}
inline void WriteVarLenUint16(BitWriter& w, uint32_t n) {
  if (n == 0) { w.put(0, 1); return; }
  w.put(1, 1);
  int nb = FloorLog2(n);
  w.put(nb, 4);
  w.put(n - (1u << nb), nb);
}

// writes a normalised distribution (sum 4096) in the codestream format
inline void WriteANSHistogram(BitWriter& w, const std::vector<int>& counts) {
  int used = 0, s0 = -1, s1 = -1;
  for (size_t i = 0; i < counts.size(); i++) if (counts[i]) { if (used == 0) s0 = (int)i; else if (used == 1) s1 = (int)i; used++; }
  if (used <= 2) {
    w.put(1, 1);  // simple
    if (used == 1) { w.put(0, 1); WriteVarLenUint8(w, s0); }
    else { w.put(1, 1); WriteVarLenUint8(w, s0); WriteVarLenUint8(w, s1); w.put(counts[s0], 12); }
    return;
  }
  w.put(0, 1); w.put(0, 1);  // not simple, not flat
  // shift = 13: unary "111" then 3 bits with (bits | 8) - 1 = 13 -> bits = 6
  w.put(1, 1); w.put(1, 1); w.put(1, 1);
  w.put(6, 3);
  const int shift = 13;
  int length = (int)counts.size();
  while (length > 0 && counts[length - 1] == 0) length--;
  if (length < 3) length = 3;
  WriteVarLenUint8(w, length - 3);
  std::vector<int> logc(length, 0);
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < length; i++) {
    int c = i < (int)counts.size() ? counts[i] : 0;
    logc[i] = c == 0 ? 0 : FloorLog2(c) + 1;
    if (logc[i] > omit_log) { omit_log = logc[i]; omit_pos = i; }
  }
  const LogCountCode& lc = LogCountCodes();
  for (int i = 0; i < length; i++) w.put(lc.bits[logc[i]], lc.nbits[logc[i]]);
  for (int i = 0; i < length; i++) {
    if (i == omit_pos || logc[i] <= 1) continue;
    int code = logc[i];
    int bitcount = std::min(std::max(0, shift - ((12 - code + 1) >> 1)), code - 1);
    int c = counts[i];
    int extra = (c - (1 << (code - 1))) >> (code - 1 - bitcount);
    if ((1 << (code - 1)) + (extra << (code - 1 - bitcount)) != c) throw std::runtime_error("count not representable");
    w.put(extra, bitcount);
  }
}

struct ClusterCode {
  std::vector<int> dist;
  std::vector<AliasE> alias;
  std::vector<std::vector<uint16_t>> reverse;
};

// prefix (Huffman) code of one cluster: code lengths (<= 15), canonical codes by (length, symbol)
struct PrefixCode { std::vector<uint8_t> len; std::vector<uint16_t> code; bool single = false; };   // single: one used symbol, coded in zero bits
// streams written from now on in this thread use prefix codes instead of ANS (what cjxl's fast efforts emit); the nested code of a
// context map stays ANS
inline bool& UsePrefixCodes() { static thread_local bool v = false; return v; }
// the Modular streams of VarDCT frames (LF coefficients, HF metadata) written from now on in this thread are LZ77-coded (what cjxl's slowest efforts may choose)
// ... their MA tree also splits on previous-channel properties (16 + 4 r + k: the sample of the r-th previous channel of equal size at this position, cjxl -E)
inline bool& UsePrevChannelProps() { static thread_local bool v = false; return v; }
inline bool& UseLz77Lf() { static thread_local bool v = false; return v; }
// ... and the AC coefficient streams of VarDCT frames
inline bool& UseLz77Ac() { static thread_local bool v = false; return v; }
// ... their MA tree has the shape cjxl writes at its default effort: the LF coefficients under a fixed tree over the weighted predictor's
// maximum error (property 15) with weighted-predictor leaves, the HF metadata under the fixed tree over the row, N and W (0: the gradient tree)
inline int& LfTreeShape() { static thread_local int v = 0; return v; }

struct EntropyCoder {
  // configuration
  bool use_prefix = false;
  std::vector<PrefixCode> prefix;    // per cluster (use_prefix)
  int log_alpha = 8;
  std::vector<uint8_t> ctx_map;      // ctx -> cluster
  std::vector<UintConfig> cfg;       // per cluster
  std::vector<ClusterCode> clusters;
  int num_ctx = 0;
  // LZ77 (dec_ans.h): symbols >= lz_min_symbol are copy lengths (hybrid uint under lz_len_cfg, + lz_min_length), each followed
  // by a distance token in the extra context num_ctx - 1
  bool lz77 = false;
  uint32_t lz_min_symbol = 224, lz_min_length = 3;
  UintConfig lz_len_cfg{4, 0, 0};
};

inline void MakePrefixCodes(EntropyCoder& ec);
inline double HistoCost(const std::vector<uint32_t>& h, uint64_t total) {
  if (total == 0) return 0;
  double c = 0;
  for (uint32_t v : h) if (v) c -= v * std::log2((double)v / (double)total);
  return c;
}

// Builds an entropy code from per-context token statistics: clusters contexts greedily (<= max_clusters) and
// normalises histograms.  tokens: all tokens that will be coded with this code.
inline void BuildEntropyCoder(const std::vector<const std::vector<Token>*>& streams, int num_ctx, const UintConfig& uc,
                              int max_clusters, EntropyCoder& ec) {
  ec.num_ctx = num_ctx;
  ec.log_alpha = 8;
  std::vector<std::vector<uint32_t>> h(num_ctx);
  std::vector<uint64_t> tot(num_ctx, 0);
  for (auto* ts : streams)
    for (const Token& t : *ts) {
      uint32_t tok, nb, bits;
      TokenSymbol(uc, t, &tok, &nb, &bits);
      if (t.ctx >= (uint32_t)num_ctx) throw std::runtime_error("token ctx out of range");
      auto& hh = h[t.ctx];
      if (hh.size() <= tok) hh.resize(tok + 1, 0);
      hh[tok]++; tot[t.ctx]++;
    }
  // greedy clustering (libjxl FastClusterHistograms-like): seed with the largest context, repeatedly add the context
  // whose cost increase vs its best cluster is largest
  std::vector<int> assign(num_ctx, -1);
  std::vector<std::vector<uint32_t>> ch;
  std::vector<uint64_t> ctot;
  std::vector<int> nonempty;
  for (int i = 0; i < num_ctx; i++) if (tot[i]) nonempty.push_back(i);
  auto dist_to = [&](int ctx, int cl) {
    // cost of coding ctx's histogram with cluster cl's distribution minus its own entropy
    const auto& a = h[ctx]; const auto& b = ch[cl];
    double c = 0;
    for (size_t s = 0; s < a.size(); s++) {
      if (!a[s]) continue;
      double p = s < b.size() && b[s] ? (double)b[s] / (double)ctot[cl] : 1.0 / 8192.0;
      c -= a[s] * std::log2(p);
    }
    return c - HistoCost(a, tot[ctx]);
  };
  if (!nonempty.empty()) {
    int seed = nonempty[0];
    for (int i : nonempty) if (tot[i] > tot[seed]) seed = i;
    ch.push_back(h[seed]); ctot.push_back(tot[seed]); assign[seed] = 0;
    std::vector<double> best(num_ctx, 0);
    for (int i : nonempty) best[i] = i == seed ? 0 : dist_to(i, 0);
    while ((int)ch.size() < max_clusters) {
      int far = -1; double fd = 0;
      for (int i : nonempty) if (assign[i] < 0 && best[i] > fd) { fd = best[i]; far = i; }
      if (far < 0 || fd < 16.0) break;
      ch.push_back(h[far]); ctot.push_back(tot[far]); assign[far] = (int)ch.size() - 1; best[far] = 0;
      int cl = (int)ch.size() - 1;
      for (int i : nonempty) if (assign[i] < 0) best[i] = std::min(best[i], dist_to(i, cl));
    }
    // final assignment against the seeds, then merge histograms
    std::vector<std::vector<uint32_t>> seeds = ch;
    std::vector<uint64_t> seedtot = ctot;
    for (int i : nonempty) {
      if (assign[i] >= 0) continue;
      int bc = 0; double bd = 1e300;
      for (int cl = 0; cl < (int)seeds.size(); cl++) {
        std::swap(ch, seeds); std::swap(ctot, seedtot);
        double d = dist_to(i, cl);
        std::swap(ch, seeds); std::swap(ctot, seedtot);
        if (d < bd) { bd = d; bc = cl; }
      }
      assign[i] = bc;
      auto& dst = ch[bc];
      if (dst.size() < h[i].size()) dst.resize(h[i].size(), 0);
      for (size_t s = 0; s < h[i].size(); s++) dst[s] += h[i][s];
      ctot[bc] += tot[i];
    }
  } else {
    ch.push_back({}); ctot.push_back(0);
  }
  // empty contexts go to cluster 0
  ec.ctx_map.assign(num_ctx, 0);
  for (int i = 0; i < num_ctx; i++) ec.ctx_map[i] = (uint8_t)(assign[i] < 0 ? 0 : assign[i]);
  ec.cfg.assign(ch.size(), uc);
  ec.clusters.resize(ch.size());
  {  // alias-table size like an encoder would pick it: smallest power of two holding the largest alphabet (>= 32)
    size_t max_alpha = 1;
    for (auto& hh : ch) { size_t n = hh.size(); while (n > 0 && hh[n - 1] == 0) n--; max_alpha = std::max(max_alpha, n); }
    ec.log_alpha = std::min(8, std::max(5, CeilLog2((uint32_t)max_alpha)));
  }
  for (size_t c = 0; c < ch.size(); c++) {
    if (ch[c].size() > 256) throw std::runtime_error("ANS alphabet > 256");
    Normalize(ch[c], ec.clusters[c].dist);
    BuildAlias(ec.clusters[c].dist, ec.log_alpha, ec.clusters[c].alias);
    // reverse map
    auto& cc = ec.clusters[c];
    cc.reverse.assign(cc.dist.size(), {});
    for (size_t s = 0; s < cc.dist.size(); s++) cc.reverse[s].assign(cc.dist[s], 0);
    const int la = ec.log_alpha, B = 4096 >> la;
    for (int v = 0; v < 4096; v++) {
      int i = v >> (12 - la), pos = v & (B - 1);
      const AliasE& e = cc.alias[i];
      bool hit = pos >= e.cutoff;
      int sym = hit ? e.right : i;
      int off = hit ? e.offs1 + pos : pos;
      if (sym >= (int)cc.reverse.size() || off >= (int)cc.reverse[sym].size()) throw std::runtime_error("alias reverse map inconsistent");
      cc.reverse[sym][off] = (uint16_t)v;
    }
  }
  if (UsePrefixCodes() && !ec.lz77) MakePrefixCodes(ec);
}

// Huffman code lengths (<= max_len) for counts (0 = unused symbol): plain Huffman, flattened (counts halved towards 1) until it fits
inline std::vector<uint8_t> HuffmanLengths(std::vector<uint32_t> counts, int max_len) {
  const size_t n = counts.size();
  std::vector<uint8_t> len(n, 0);
  size_t used = 0;
  for (uint32_t c : counts) used += c != 0;
  if (used == 0) return len;
  if (used == 1) { for (size_t i = 0; i < n; i++) if (counts[i]) len[i] = 1; return len; }
  for (;;) {
    struct Node { uint64_t w; int l, r; };
    std::vector<Node> nodes;
    std::vector<int> live;
    for (size_t i = 0; i < n; i++) if (counts[i]) { nodes.push_back({counts[i], -1, (int)i}); live.push_back((int)nodes.size() - 1); }
    while (live.size() > 1) {
      std::sort(live.begin(), live.end(), [&](int a, int b) { return nodes[a].w > nodes[b].w || (nodes[a].w == nodes[b].w && a < b); });
      const int a = live.back(); live.pop_back();
      const int b = live.back(); live.pop_back();
      nodes.push_back({nodes[a].w + nodes[b].w, a, b});
      live.push_back((int)nodes.size() - 1);
    }
    std::fill(len.begin(), len.end(), 0);
    int deepest = 0;
    std::vector<std::pair<int, int>> stack{{live[0], 0}};
    while (!stack.empty()) {
      auto [id, d] = stack.back(); stack.pop_back();
      if (nodes[id].l < 0) { len[nodes[id].r] = (uint8_t)d; deepest = std::max(deepest, d); }
      else { stack.push_back({nodes[id].l, d + 1}); stack.push_back({nodes[id].r, d + 1}); }
    }
    if (deepest <= max_len) return len;
    for (auto& c : counts) if (c) c = c / 2 + 1;
  }
}
inline std::vector<uint16_t> CanonicalCodes(const std::vector<uint8_t>& len) {
  std::vector<uint16_t> code(len.size(), 0);
  uint32_t next = 0;
  for (int l = 1; l <= 15; l++) {
    for (size_t s = 0; s < len.size(); s++) if (len[s] == l) code[s] = (uint16_t)next++;
    next <<= 1;
  }
  return code;
}
inline void PutCodeMsbFirst(BitWriter& w, uint32_t code, int len) { for (int b = len - 1; b >= 0; b--) w.put((code >> b) & 1, 1); }
// Brotli-style description of a prefix code over `alphabet` symbols (dec_huffman.cc ReadHuffmanCode): the simple form for one or two
// used symbols, else the complex form — code-length code over the lengths that occur (no repeat symbols), then one length per symbol up
// to the last used one.
inline void WritePrefixCodeDescription(BitWriter& w, const std::vector<uint8_t>& len, int alphabet) {
  if (alphabet == 1) return;
  std::vector<int> used;
  for (int i = 0; i < alphabet; i++) if (len[i]) used.push_back(i);
  int max_bits = 0;
  for (int v = alphabet - 1; v; v >>= 1) max_bits++;
  if (used.size() <= 2) {
    w.put(1, 2);                                  // simple
    w.put((uint32_t)std::max<size_t>(used.size(), 1) - 1, 2);
    if (used.empty()) w.put(0, max_bits);
    for (int sym : used) w.put((uint32_t)sym, max_bits);
    return;
  }
  w.put(0, 2);                                    // complex, no skipped code-length-code entries
  std::vector<uint32_t> freq(18, 0);
  for (int i = 0; i <= used.back(); i++) freq[len[i]]++;
  std::vector<uint8_t> cl = HuffmanLengths(freq, 5);
  int distinct = 0;
  for (int i = 0; i < 18; i++) distinct += cl[i] != 0;
  static const uint8_t kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  // fixed code of the code-length-code lengths: 0 -> 00, 4 -> 10 (first bit 1), 3 -> 01, 2 -> 110, 1 -> 1110, 5 -> 1111 (bits in stream order)
  auto put_cl = [&](int v) {
    switch (v) { case 0: w.put(0, 2); break; case 4: w.put(1, 2); break; case 3: w.put(2, 2); break; case 2: w.put(3, 3); break; case 1: w.put(7, 4); break; default: w.put(15, 4); break; }
  };
  int space = 32;
  for (int i = 0; i < 18 && space > 0; i++) {
    const int v = cl[kOrder[i]];
    put_cl(v);
    if (v) space -= 32 >> v;
  }
  if (distinct != 1 && space != 0) throw std::runtime_error("prefix writer: incomplete code-length code");
  if (distinct == 1) return;                      // every symbol up to the end has that one length: nothing more is read... (only valid if complete)
  const std::vector<uint16_t> clcode = CanonicalCodes(cl);
  int sp = 32768;
  for (int i = 0; i <= used.back() && sp > 0; i++) {
    PutCodeMsbFirst(w, clcode[len[i]], cl[len[i]]);
    if (len[i]) sp -= 32768 >> len[i];
  }
  if (sp != 0) throw std::runtime_error("prefix writer: incomplete code");
}
inline void MakePrefixCodes(EntropyCoder& ec) {
  ec.use_prefix = true;
  ec.log_alpha = 15;
  ec.prefix.resize(ec.clusters.size());
  for (size_t c = 0; c < ec.clusters.size(); c++) {
    std::vector<uint32_t> counts(ec.clusters[c].dist.begin(), ec.clusters[c].dist.end());
    while (!counts.empty() && counts.back() == 0) counts.pop_back();
    if (counts.empty()) counts.push_back(1);
    ec.prefix[c].len = HuffmanLengths(counts, 15);
    size_t used = 0;
    for (uint8_t l : ec.prefix[c].len) used += l != 0;
    ec.prefix[c].single = used <= 1;                            // a single symbol costs no bits (ReadSymbol: count[0] flags it)
    ec.prefix[c].code = CanonicalCodes(ec.prefix[c].len);
  }
}

inline void WriteUintConfig(BitWriter& w, const UintConfig& c, int log_alpha) {
  w.put(c.split_exponent, CeilLog2(log_alpha + 1));
  if (c.split_exponent != log_alpha) {
    w.put(c.msb, CeilLog2(c.split_exponent + 1));
    w.put(c.lsb, CeilLog2(c.split_exponent - c.msb + 1));
  }
}

inline void WriteEntropyCode(BitWriter& w, const EntropyCoder& ec);
inline void EncodeTokens(BitWriter& w, const EntropyCoder& ec, const std::vector<Token>& tokens);

// context map (dec_context_map.cc counterpart)
inline void WriteContextMap(BitWriter& w, const std::vector<uint8_t>& map, int num_clusters) {
  int bits = num_clusters <= 1 ? 0 : CeilLog2(num_clusters);
  if (bits <= 3 && map.size() * bits <= 2048) {
    w.put(1, 1);  // simple
    w.put(bits, 2);
    for (uint8_t m : map) w.put(m, bits);
    return;
  }
  w.put(0, 1);
  // move-to-front transform
  std::vector<Token> toks;
  uint8_t mtf[256];
  for (int i = 0; i < 256; i++) mtf[i] = (uint8_t)i;
  for (uint8_t m : map) {
    int idx = 0;
    while (mtf[idx] != m) idx++;
    toks.push_back({0, (uint32_t)idx});
    for (int j = idx; j > 0; j--) mtf[j] = mtf[j - 1];
    mtf[0] = m;
  }
  w.put(1, 1);  // use_mtf
  EntropyCoder nested;
  std::vector<const std::vector<Token>*> ss{&toks};
  const bool outer_prefix = UsePrefixCodes();
  UsePrefixCodes() = false;
  BuildEntropyCoder(ss, 1, UintConfig{4, 2, 0}, 1, nested);
  UsePrefixCodes() = outer_prefix;
  WriteEntropyCode(w, nested);
  EncodeTokens(w, nested, toks);
}

inline void WriteEntropyCode(BitWriter& w, const EntropyCoder& ec) {
  if (!ec.lz77) w.put(0, 1);
  else {
    w.put(1, 1);
    WriteU32(w, ec.lz_min_symbol, {0, 224}, {0, 512}, {0, 4096}, {15, 8});
    WriteU32(w, ec.lz_min_length, {0, 3}, {0, 4}, {2, 5}, {8, 9});
    WriteUintConfig(w, ec.lz_len_cfg, 8);
  }
  if (ec.num_ctx > 1) WriteContextMap(w, ec.ctx_map, (int)ec.clusters.size());
  if (ec.use_prefix) {
    w.put(1, 1);  // prefix codes: log_alpha_size is 15
    for (size_t c = 0; c < ec.clusters.size(); c++) WriteUintConfig(w, ec.cfg[c], 15);
    for (size_t c = 0; c < ec.clusters.size(); c++) WriteVarLenUint16(w, (uint32_t)ec.prefix[c].len.size() - 1);
    for (size_t c = 0; c < ec.clusters.size(); c++) WritePrefixCodeDescription(w, ec.prefix[c].len, (int)ec.prefix[c].len.size());
    return;
  }
  w.put(0, 1);  // ANS (no prefix codes)
  w.put(ec.log_alpha - 5, 2);
  for (size_t c = 0; c < ec.clusters.size(); c++) WriteUintConfig(w, ec.cfg[c], ec.log_alpha);
  for (size_t c = 0; c < ec.clusters.size(); c++) WriteANSHistogram(w, ec.clusters[c].dist);
}

// rANS encode one stream: [state32] then per token (refill16?) + extra bits, in decode order
inline void EncodeTokens(BitWriter& w, const EntropyCoder& ec, const std::vector<Token>& tokens) {
  const size_t n = tokens.size();
  if (ec.use_prefix) {   // no state: per token its code (first bit = most significant), then the extra bits
    for (size_t i = 0; i < n; i++) {
      const Token& t = tokens[i];
      const int cl = ec.ctx_map[t.ctx];
      uint32_t tok, nb, bits;
      TokenSymbol(ec.cfg[cl], t, &tok, &nb, &bits);
      const PrefixCode& pc = ec.prefix[cl];
      if (tok >= pc.len.size()) throw std::runtime_error("prefix: symbol outside the alphabet");
      if (!pc.single) { if (!pc.len[tok]) throw std::runtime_error("prefix: symbol without a code"); PutCodeMsbFirst(w, pc.code[tok], pc.len[tok]); }
      if (nb > 24) { w.put(bits & 0xFFFF, 16); w.put(bits >> 16, nb - 16); } else w.put(bits, nb);
    }
    return;
  }
  std::vector<uint16_t> refill(n, 0);
  std::vector<uint8_t> has_refill(n, 0);
  uint32_t state = 0x130000;
  for (size_t ii = n; ii-- > 0;) {
    const Token& t = tokens[ii];
    int cl = ec.ctx_map[t.ctx];
    uint32_t tok, nb, bits;
    TokenSymbol(ec.cfg[cl], t, &tok, &nb, &bits);
    const ClusterCode& cc = ec.clusters[cl];
    if (tok >= cc.dist.size() || cc.dist[tok] == 0) throw std::runtime_error("symbol with zero probability");
    uint32_t freq = cc.dist[tok];
    if ((state >> 20) >= freq) { refill[ii] = (uint16_t)(state & 0xFFFF); has_refill[ii] = 1; state >>= 16; }
    state = ((state / freq) << 12) + cc.reverse[tok][state % freq];
  }
  w.put(state & 0xFFFF, 16); w.put(state >> 16, 16);
  for (size_t i = 0; i < n; i++) {
    const Token& t = tokens[i];
    int cl = ec.ctx_map[t.ctx];
    uint32_t tok, nb, bits;
    TokenSymbol(ec.cfg[cl], t, &tok, &nb, &bits);
    if (has_refill[i]) w.put(refill[i], 16);
    if (nb > 24) { w.put(bits & 0xFFFF, 16); w.put(bits >> 16, nb - 16); } else w.put(bits, nb);
  }
}

}  // namespace synth
