// jxl-hip: host-side header parser (see host_parse.h).  Field layouts: SURVEY.md App. B.1-B.6.
#include "host_parse.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <mutex>

namespace jxlhip {

namespace {

[[noreturn]] void Fail(const char* msg) { throw ParseError(msg, false); }
[[noreturn]] void Unsupported(const char* msg) { throw ParseError(std::string("unsupported: ") + msg, true); }

struct Reader {
  BitReader br;
  uint64_t limit_bits;
  Reader(const Codestream& cs, uint64_t bitpos) {
    br.Init(cs.data(), bitpos, cs.size);
    limit_bits = cs.size * 8;
  }
  uint32_t u(int n) {
    uint32_t v = n ? br.Read(n) : 0;
    if (br.BitPos() > limit_bits) throw ParseError("truncated", false);
    return v;
  }
  bool b() { return u(1) != 0; }
  uint64_t pos() const { return br.BitPos(); }
  void align() { int r = (int)(pos() & 7); if (r) u(8 - r); }
  struct D { int bits; uint32_t off; };
  uint32_t U32(D d0, D d1, D d2, D d3) {
    uint32_t s = u(2);
    D d = s == 0 ? d0 : s == 1 ? d1 : s == 2 ? d2 : d3;
    return d.off + u(d.bits);
  }
  uint64_t U64() {
    uint32_t s = u(2);
    if (s == 0) return 0;
    if (s == 1) return 1 + u(4);
    if (s == 2) return 17 + u(8);
    uint64_t v = u(12);
    int shift = 12;
    while (u(1)) {
      if (shift == 60) { v |= (uint64_t)u(4) << shift; break; }
      v |= (uint64_t)u(8) << shift;
      shift += 8;
    }
    return v;
  }
  float F16() {
    uint32_t h = u(16);
    uint32_t sign = h >> 15, e = (h >> 10) & 31, m = h & 1023;
    if (e == 31) Fail("F16 is inf/nan");
    float v = e == 0 ? std::ldexp((float)m, -24) : std::ldexp((float)(m + 1024), (int)e - 25);
    return sign ? -v : v;
  }
  uint32_t Enum() { return U32({0, 0}, {0, 1}, {4, 2}, {6, 18}); }
  void SkipExtensions() {
    uint64_t ext = U64();
    if (!ext) return;
    uint64_t total = 0;
    for (int i = 0; i < 64; i++) if (ext >> i & 1) total += U64();
    while (total > 0) { int n = (int)std::min<uint64_t>(total, 32); u(n); total -= n; }
  }
};

int CeilLog2(uint32_t x) { int r = 0; while ((1ull << r) < x) r++; return r; }
int FloorLog2(uint32_t x) { int r = 0; while (x >>= 1) r++; return r; }

// ---- entropy code header -----------------------------------------------------------------------------------------
uint32_t VarLenUint8(Reader& r) { if (!r.u(1)) return 0; int n = r.u(3); return n ? r.u(n) + (1u << n) : 1; }
uint32_t VarLenUint16(Reader& r) { if (!r.u(1)) return 0; int n = r.u(4); return n ? r.u(n) + (1u << n) : 1; }

uint32_t ReadUintConfig(Reader& r, int log_alpha) {
  uint32_t split = r.u(CeilLog2(log_alpha + 1)), msb = 0, lsb = 0;
  if (split != (uint32_t)log_alpha) {
    msb = r.u(CeilLog2(split + 1));
    if (msb > split) Fail("hybrid uint: msb_in_token");
    lsb = r.u(CeilLog2(split - msb + 1));
  }
  if (lsb + msb > split) Fail("hybrid uint: lsb_in_token");
  return split | (msb << 8) | (lsb << 16);
}

void ReadHistogram(Reader& r, vec<int>& counts) {
  counts.clear();
  if (r.u(1)) {
    int ns = r.u(1) + 1;
    uint32_t s0 = VarLenUint8(r);
    if (ns == 1) { counts.assign(s0 + 1, 0); counts[s0] = 4096; return; }
    uint32_t s1 = VarLenUint8(r);
    if (s0 == s1) Fail("ANS: duplicate symbol");
    counts.assign(std::max(s0, s1) + 1, 0);
    counts[s0] = r.u(12);
    counts[s1] = 4096 - counts[s0];
    return;
  }
  if (r.u(1)) {
    int n = VarLenUint8(r) + 1;
    counts.assign(n, 4096 / n);
    for (int i = 0; i < 4096 % n; i++) counts[i]++;
    return;
  }
  int len = 0;
  while (len < 3 && r.u(1)) len++;
  int shift = (int)(r.u(len) | (1u << len)) - 1;
  if (shift > 13) Fail("ANS: shift");
  int length = VarLenUint8(r) + 3;
  // 7-bit peek LUT of the log-count prefix code: entries {nbits, value}
  static const uint8_t base[16][2] = {{3, 10}, {7, 12}, {3, 7}, {4, 3}, {3, 6}, {3, 8}, {3, 9}, {4, 5}, {3, 10}, {4, 4}, {3, 7}, {4, 1}, {3, 6}, {3, 8}, {3, 9}, {4, 2}};
  static const uint8_t odd[8][2] = {{7, 12}, {5, 0}, {6, 11}, {5, 0}, {7, 13}, {5, 0}, {6, 11}, {5, 0}};
  vec<int> logc(length, 0), same(length, 0);
  int omit_log = -1, omit_pos = -1;
  for (int i = 0; i < length; i++) {
    r.br.Refill();
    uint32_t idx = r.br.Peek(7);
    const uint8_t* e = (idx & 15) == 1 ? odd[idx >> 4] : base[idx & 15];
    r.u(e[0]);
    logc[i] = e[1];
    if (logc[i] == 13) {
      int rl = VarLenUint8(r);
      same[i] = rl + 5;
      i += rl + 3;
      continue;
    }
    if (logc[i] > omit_log) { omit_log = logc[i]; omit_pos = i; }
  }
  if (omit_pos < 0) Fail("ANS: no omit position");
  if (omit_pos + 1 < length && logc[omit_pos + 1] == 13) Fail("ANS: RLE after omit position");
  counts.assign(length, 0);
  int total = 0, prev = 0, numsame = 0;
  for (int i = 0; i < length; i++) {
    if (same[i]) { numsame = same[i] - 1; prev = i > 0 ? counts[i - 1] : 0; }
    if (numsame > 0) { counts[i] = prev; numsame--; }
    else {
      int code = logc[i];
      if (i == omit_pos || code == 0) continue;
      if (code == 1) counts[i] = 1;
      else {
        int bitcount = std::min(std::max(0, shift - ((12 - code + 1) >> 1)), code - 1);
        counts[i] = (1 << (code - 1)) + (r.u(bitcount) << (code - 1 - bitcount));
      }
    }
    total += counts[i];
  }
  counts[omit_pos] = 4096 - total;
  if (counts[omit_pos] <= 0) Fail("ANS: omitted count <= 0");
}

void BuildAlias(vec<int> dist, int log_alpha, uint64_t* out) {
  const int T = 1 << log_alpha, B = 4096 >> log_alpha;
  while (!dist.empty() && dist.back() == 0) dist.pop_back();
  if (dist.empty()) dist.assign(1, 4096);
  if ((int)dist.size() > T) Fail("ANS: alphabet larger than table");
  for (size_t s = 0; s < dist.size(); s++) {
    if (dist[s] == 4096) {
      for (int i = 0; i < T; i++) out[i] = PackAlias(0, (uint32_t)s, 0, (uint32_t)(B * i), 4096);
      return;
    }
  }
  vec<int> cut(T, 0), right(T, 0), offs1(T, 0), over, under;
  for (size_t i = 0; i < dist.size(); i++) cut[i] = dist[i];
  for (int i = 0; i < T; i++) { if (cut[i] > B) over.push_back(i); else if (cut[i] < B) under.push_back(i); }
  while (!over.empty()) {
    if (under.empty()) Fail("ANS: alias construction");
    int o = over.back(); over.pop_back();
    int u = under.back(); under.pop_back();
    cut[o] -= B - cut[u];
    right[u] = o; offs1[u] = cut[o];
    if (cut[o] < B) under.push_back(o); else if (cut[o] > B) over.push_back(o);
  }
  for (int i = 0; i < T; i++) {
    if (cut[i] == B) { right[i] = i; offs1[i] = 0; cut[i] = 0; } else offs1[i] -= cut[i];
    uint32_t f0 = i < (int)dist.size() ? dist[i] : 0, f1 = right[i] < (int)dist.size() ? dist[right[i]] : 0;
    out[i] = PackAlias((uint32_t)cut[i], (uint32_t)right[i], f0, (uint32_t)offs1[i], f1);
  }
}

// Brotli-style prefix code → canonical (count per length, symbols sorted by (length, value))
void ReadPrefixCode(Reader& r, int alphabet, uint16_t* count16, vec<uint16_t>& syms) {
  for (int i = 0; i < 16; i++) count16[i] = 0;
  vec<uint8_t> lens(alphabet, 0);
  auto finish = [&]() {
    int nonzero = 0, last = 0;
    for (int i = 0; i < alphabet; i++) if (lens[i]) { nonzero++; last = i; }
    if (nonzero <= 1) { count16[0] = 1; syms.push_back((uint16_t)(nonzero ? last : 0)); return; }
    for (int len = 1; len <= 15; len++) for (int i = 0; i < alphabet; i++) if (lens[i] == len) { count16[len]++; syms.push_back((uint16_t)i); }
  };
  if (alphabet == 1) { finish(); return; }
  int hskip = r.u(2);
  if (hskip == 1) {
    int max_bits = 0;
    for (int v = alphabet - 1; v; v >>= 1) max_bits++;
    int n = r.u(2) + 1, s[4];
    for (int i = 0; i < n; i++) { s[i] = r.u(max_bits); if (s[i] >= alphabet) Fail("prefix: symbol out of range"); }
    for (int i = 0; i < n; i++) for (int j = i + 1; j < n; j++) if (s[i] == s[j]) Fail("prefix: duplicate symbol");
    if (n == 1) { count16[0] = 1; syms.push_back((uint16_t)s[0]); return; }
    if (n == 2) { lens[s[0]] = 1; lens[s[1]] = 1; }
    else if (n == 3) { lens[s[0]] = 1; lens[s[1]] = 2; lens[s[2]] = 2; }
    else if (r.u(1)) { lens[s[0]] = 1; lens[s[1]] = 2; lens[s[2]] = 3; lens[s[3]] = 3; }
    else for (int i = 0; i < 4; i++) lens[s[i]] = 2;
    finish();
    return;
  }
  static const uint8_t kOrder[18] = {1, 2, 3, 4, 0, 5, 17, 6, 16, 7, 8, 9, 10, 11, 12, 13, 14, 15};
  static const uint8_t kLen[16] = {2, 2, 2, 3, 2, 2, 2, 4, 2, 2, 2, 3, 2, 2, 2, 4};
  static const uint8_t kVal[16] = {0, 4, 3, 2, 0, 4, 3, 1, 0, 4, 3, 2, 0, 4, 3, 5};
  uint8_t cl[18] = {0};
  int space = 32, ncodes = 0;
  for (int i = hskip; i < 18 && space > 0; i++) {
    r.br.Refill();
    uint32_t p = r.br.Peek(4);
    r.u(kLen[p]);
    cl[kOrder[i]] = kVal[p];
    if (kVal[p]) { space -= 32 >> kVal[p]; ncodes++; }
  }
  if (ncodes != 1 && space != 0) Fail("prefix: code length code");
  // canonical decode table for the code-length code (max length 5)
  int cl_single = -1;
  if (ncodes == 1) for (int i = 0; i < 18; i++) if (cl[i]) cl_single = i;
  uint8_t lut_sym[32], lut_len[32];
  if (cl_single < 0) {
    uint32_t code = 0;
    for (int len = 1; len <= 5; len++) {
      for (int s = 0; s < 18; s++) {
        if (cl[s] != len) continue;
        uint32_t rev = 0;
        for (int b = 0; b < len; b++) if (code >> b & 1) rev |= 1u << (len - 1 - b);
        for (uint32_t k = rev; k < 32; k += 1u << len) { lut_sym[k] = (uint8_t)s; lut_len[k] = (uint8_t)len; }
        code++;
      }
      code <<= 1;
    }
  }
  int symbol = 0, prev_len = 8, repeat = 0, repeat_len = 0, sp = 32768;
  while (symbol < alphabet && sp > 0) {
    int v;
    if (cl_single >= 0) v = cl_single;
    else { r.br.Refill(); uint32_t p = r.br.Peek(5); v = lut_sym[p]; r.u(lut_len[p]); }
    if (v < 16) {
      repeat = 0;
      lens[symbol++] = (uint8_t)v;
      if (v) { prev_len = v; sp -= 32768 >> v; }
    } else {
      int extra = v == 16 ? 2 : 3, new_len = v == 16 ? prev_len : 0;
      if (repeat_len != new_len) { repeat = 0; repeat_len = new_len; }
      int old = repeat;
      if (repeat > 0) { repeat -= 2; repeat <<= extra; }
      repeat += r.u(extra) + 3;
      int delta = repeat - old;
      if (symbol + delta > alphabet) Fail("prefix: repeat overflow");
      for (int i = 0; i < delta; i++) lens[symbol++] = (uint8_t)repeat_len;
      if (repeat_len) sp -= delta << (15 - repeat_len);
    }
  }
  if (sp != 0) Fail("prefix: incomplete code");
  finish();
}

void ReadEntropyCode(Reader& r, uint32_t num_ctx, HostCode* hc, bool allow_lz77 = true);

struct HostSymbolReader {  // symbol reader over a HostCode, for the small global streams parsed on the host
  Reader& r; const HostCode& hc; DevCode view; AnsReader ans;
  vec<uint32_t> window; Lz77State lz;
  HostSymbolReader(Reader& rr, const HostCode& c, uint32_t dist_multiplier = 0) : r(rr), hc(c), view(c.View()) {
    ans.Init(r.br, view);
    if (hc.lz77) { window.assign(Lz77State::kWindow, 0); lz.Init(window.data(), dist_multiplier); }
  }
  uint32_t Read(uint32_t ctx) {
    uint32_t v;
    if (!hc.lz77) v = ReadHybridUint(r.br, ans, view, ctx);
    else v = Lz77Read(r.br, lz, ctx, hc.num_ctx, hc.lz_min_symbol, hc.lz_min_length, hc.lz_len_cfg,
                      [&](uint32_t c) { return (uint32_t)view.ctx_map[c]; }, [&](uint32_t cl) { return ReadSymbol(r.br, ans, view, cl); },
                      [&](uint32_t cl) { return view.cfg[cl]; });
    if (r.pos() > r.limit_bits) throw ParseError("truncated", false);
    return v;
  }
  void CheckFinal() { if (!ans.FinalOk(view)) Fail("ANS final state"); }
};

void ReadContextMap(Reader& r, uint32_t num_ctx, vec<uint8_t>& map, uint32_t* num_clusters) {
  map.assign(num_ctx, 0);
  if (r.b()) {
    int bits = r.u(2);
    for (auto& m : map) m = (uint8_t)r.u(bits);
  } else {
    bool mtf = r.b();
    HostCode nested;
    ReadEntropyCode(r, 1, &nested, num_ctx > 2);
    HostSymbolReader sr(r, nested);
    for (auto& m : map) { uint32_t v = sr.Read(0); if (v > 255) Fail("context map value"); m = (uint8_t)v; }
    sr.CheckFinal();
    if (mtf) {
      uint8_t t[256];
      for (int i = 0; i < 256; i++) t[i] = (uint8_t)i;
      for (auto& m : map) {
        uint8_t idx = m, v = t[idx];
        m = v;
        for (int j = idx; j > 0; j--) t[j] = t[j - 1];
        t[0] = v;
      }
    }
  }
  uint32_t mx = 0;
  for (auto m : map) mx = std::max<uint32_t>(mx, m);
  vec<bool> used(mx + 1, false);
  for (auto m : map) used[m] = true;
  for (bool u : used) if (!u) Fail("context map skips a cluster");
  *num_clusters = mx + 1;
}

void ReadEntropyCode(Reader& r, uint32_t num_ctx, HostCode* hc, bool allow_lz77) {
  *hc = HostCode();
  hc->lz77 = r.b();
  if (hc->lz77) {
    if (!allow_lz77) Fail("LZ77 not allowed");
    hc->lz_min_symbol = r.U32({0, 224}, {0, 512}, {0, 4096}, {15, 8});
    hc->lz_min_length = r.U32({0, 3}, {0, 4}, {2, 5}, {8, 9});
    hc->lz_len_cfg = ReadUintConfig(r, 8);
  }
  hc->num_ctx = num_ctx;
  if (hc->lz77) num_ctx += 1;            // the distance context (last entry of the context map)
  if (num_ctx > 1) ReadContextMap(r, num_ctx, hc->ctx_map, &hc->num_clusters);
  else { hc->ctx_map.assign(1, 0); hc->num_clusters = 1; }
  hc->use_prefix = r.b();
  hc->log_alpha = hc->use_prefix ? 15 : 5 + r.u(2);
  hc->cfg.resize(hc->num_clusters);
  for (auto& c : hc->cfg) c = ReadUintConfig(r, hc->log_alpha);
  if (hc->use_prefix) {
    vec<int> asz(hc->num_clusters);
    for (auto& a : asz) { a = VarLenUint16(r) + 1; if (a > (1 << 15)) Fail("prefix alphabet too large"); }
    hc->pfx_count.assign(hc->num_clusters * 16, 0);
    hc->pfx_sym_off.resize(hc->num_clusters);
    for (uint32_t c = 0; c < hc->num_clusters; c++) {
      hc->pfx_sym_off[c] = (uint32_t)hc->pfx_syms.size();
      ReadPrefixCode(r, asz[c], &hc->pfx_count[c * 16], hc->pfx_syms);
    }
    hc->alias.assign(1, 0);
  } else {
    hc->alias.assign((size_t)hc->num_clusters << hc->log_alpha, 0);
    vec<int> dist;
    for (uint32_t c = 0; c < hc->num_clusters; c++) {
      ReadHistogram(r, dist);
      BuildAlias(dist, hc->log_alpha, &hc->alias[(size_t)c << hc->log_alpha]);
    }
    hc->pfx_count.assign(16, 0); hc->pfx_sym_off.assign(1, 0); hc->pfx_syms.assign(1, 0);
  }
}

// ---- embedded ICC profile (icc_codec.cc ICCReader + UnpredictICC) ----------------------------------------------------------------------
// The codestream carries the profile as an entropy-coded byte stream (41 contexts chosen from the two previous bytes) of a
// *predicted* form: varint output size, varint command-stream size, commands, then data.  The 128-byte header is coded as
// differences from a template, the tag table as one command per tag, the tag contents as insert / shuffle / N-th order
// prediction runs.
namespace {
uint32_t IccContext(size_t i, uint32_t b1, uint32_t b2) {
  if (i <= 128) return 0;
  auto letter = [](uint32_t b) { return (b >= 'a' && b <= 'z') || (b >= 'A' && b <= 'Z'); };
  auto digit = [](uint32_t b) { return (b >= '0' && b <= '9') || b == '.' || b == ','; };
  uint32_t k1, k2;
  if (letter(b1)) k1 = 0; else if (digit(b1)) k1 = 1; else if (b1 <= 1) k1 = 2 + b1; else if (b1 < 16) k1 = 4; else if (b1 > 240 && b1 < 255) k1 = 5; else if (b1 == 255) k1 = 6; else k1 = 7;
  if (letter(b2)) k2 = 0; else if (digit(b2)) k2 = 1; else if (b2 < 16) k2 = 2; else if (b2 > 240) k2 = 3; else k2 = 4;
  return 1 + k1 + k2 * 8;
}
struct IccIn {
  const vec<uint8_t>& e; size_t size;
  uint64_t VarInt(size_t* pos) const {
    uint64_t v = 0; int shift = 0;
    for (;;) {
      if (*pos >= size || shift > 63) Fail("ICC varint");
      const uint8_t b = e[(*pos)++];
      v |= (uint64_t)(b & 127) << shift;
      if (!(b & 128)) return v;
      shift += 7;
    }
  }
};
void IccShuffle(uint8_t* data, size_t size, size_t width) {
  const size_t height = (size + width - 1) / width;
  vec<uint8_t> out(size);
  size_t s = 0, j = 0;
  for (size_t i = 0; i < size; i++) { out[i] = data[j]; j += height; if (j >= size) j = ++s; }
  memcpy(data, out.data(), size);
}
void Put32(vec<uint8_t>& v, uint64_t x) { v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x); }
void PutTag(vec<uint8_t>& v, const char* t) { for (int i = 0; i < 4; i++) v.push_back((uint8_t)t[i]); }
uint8_t IccPredict(const vec<uint8_t>& d, size_t start, size_t i, size_t stride, size_t width, int order) {
  auto pr = [&](uint64_t p1, uint64_t p2, uint64_t p3) -> uint64_t { return order == 0 ? p1 : order == 1 ? 2 * p1 - p2 : 3 * p1 - 3 * p2 + p3; };
  if (width == 1) { const size_t pos = start + i; return (uint8_t)pr(d[pos - stride], d[pos - stride * 2], d[pos - stride * 3]); }
  if (width == 2) {
    const size_t p = start + (i & ~(size_t)1);
    auto rd = [&](size_t o) -> uint64_t { return ((uint64_t)d[o] << 8) | d[o + 1]; };
    const uint64_t v = pr(rd(p - stride), rd(p - stride * 2), rd(p - stride * 3)) & 0xFFFF;
    return (i & 1) ? (uint8_t)(v & 255) : (uint8_t)(v >> 8);
  }
  const size_t p = start + (i & ~(size_t)3);
  auto rd = [&](size_t o) -> uint64_t { return ((uint64_t)d[o] << 24) | ((uint64_t)d[o + 1] << 16) | ((uint64_t)d[o + 2] << 8) | d[o + 3]; };
  const uint64_t v = pr(rd(p - stride), rd(p - stride * 2), rd(p - stride * 3)) & 0xFFFFFFFFull;
  return (uint8_t)(v >> ((3 - (i & 3)) * 8));
}
}  // namespace

void UnpredictIcc(const vec<uint8_t>& enc, vec<uint8_t>* out) {
  const IccIn in{enc, enc.size()};
  const size_t size = enc.size();
  vec<uint8_t>& res = *out;
  res.clear();
  size_t pos = 0;
  const uint64_t osize = in.VarInt(&pos);
  if (osize >> 32) Fail("ICC size");
  const uint64_t csize = in.VarInt(&pos);
  if (csize >> 32) Fail("ICC command size");
  size_t cpos = pos;
  if (csize > size - pos) Fail("ICC command stream out of bounds");
  const size_t cend = cpos + csize;
  pos = cend;
  // header: differences from the predicted header (template + profile size; a few fields copy earlier ones)
  uint8_t header[128] = {0};
  header[8] = 4;
  memcpy(header + 12, "mntr", 4); memcpy(header + 16, "RGB ", 4); memcpy(header + 20, "XYZ ", 4); memcpy(header + 36, "acsp", 4);
  header[70] = 246; header[71] = 214; header[73] = 1; header[78] = 211; header[79] = 45;
  header[0] = (uint8_t)(osize >> 24); header[1] = (uint8_t)(osize >> 16); header[2] = (uint8_t)(osize >> 8); header[3] = (uint8_t)osize;
  for (size_t i = 0; i <= 128; i++) {
    if (res.size() == osize) { if (cpos != cend || pos != size) Fail("ICC: data left after the profile"); return; }
    if (i == 128) break;
    if (i == 8 && res.size() >= 8) { header[80] = res[4]; header[81] = res[5]; header[82] = res[6]; header[83] = res[7]; }
    if (i == 41 && res.size() >= 41) {
      if (res[40] == 'A') { header[41] = 'P'; header[42] = 'P'; header[43] = 'L'; }
      if (res[40] == 'M') { header[41] = 'S'; header[42] = 'F'; header[43] = 'T'; }
    }
    if (i == 42 && res.size() >= 42) {
      if (res[40] == 'S' && res[41] == 'G') { header[42] = 'I'; header[43] = ' '; }
      if (res[40] == 'S' && res[41] == 'U') { header[42] = 'N'; header[43] = 'W'; }
    }
    if (pos >= size) Fail("ICC header out of bounds");
    res.push_back((uint8_t)(enc[pos++] + header[i]));
  }
  if (cpos >= cend) Fail("ICC commands out of bounds");
  // tag table
  static const char* const kTagStrings[17] = {"cprt", "wtpt", "bkpt", "rXYZ", "gXYZ", "bXYZ", "kXYZ", "rTRC", "gTRC", "bTRC", "kTRC", "chad", "desc", "chrm", "dmnd", "dmdd", "lumi"};
  uint64_t numtags = in.VarInt(&cpos);
  if (numtags != 0) {
    numtags--;
    if (numtags >> 32) Fail("ICC tag count");
    Put32(res, numtags);
    uint64_t prevtagstart = 128 + numtags * 12, prevtagsize = 0;
    for (;;) {
      if (res.size() > osize || cpos > cend) Fail("ICC tag list out of bounds");
      if (cpos == cend) break;
      const uint8_t command = enc[cpos++];
      const uint8_t tagcode = command & 63;
      char tag[4];
      if (tagcode == 0) break;
      else if (tagcode == 1) { if (size - pos < 4) Fail("ICC tag out of bounds"); memcpy(tag, &enc[pos], 4); pos += 4; }
      else if (tagcode == 2) memcpy(tag, "rTRC", 4);
      else if (tagcode == 3) memcpy(tag, "rXYZ", 4);
      else { if (tagcode - 4 >= 17) Fail("ICC tag code"); memcpy(tag, kTagStrings[tagcode - 4], 4); }
      for (int i = 0; i < 4; i++) res.push_back((uint8_t)tag[i]);
      uint64_t tagstart, tagsize = prevtagsize;
      auto is = [&](const char* t) { return memcmp(tag, t, 4) == 0; };
      if (is("rXYZ") || is("gXYZ") || is("bXYZ") || is("kXYZ") || is("wtpt") || is("bkpt") || is("lumi")) tagsize = 20;
      if (command & 64) { if (cpos >= cend) Fail("ICC commands out of bounds"); tagstart = in.VarInt(&cpos); }
      else tagstart = prevtagstart + prevtagsize;
      if (tagstart >> 32) Fail("ICC tag offset");
      Put32(res, tagstart);
      if (command & 128) { if (cpos >= cend) Fail("ICC commands out of bounds"); tagsize = in.VarInt(&cpos); }
      if (tagsize >> 32) Fail("ICC tag size");
      Put32(res, tagsize);
      prevtagstart = tagstart; prevtagsize = tagsize;
      if (tagcode == 2) { PutTag(res, "gTRC"); Put32(res, tagstart); Put32(res, tagsize); PutTag(res, "bTRC"); Put32(res, tagstart); Put32(res, tagsize); }
      if (tagcode == 3) {
        if ((tagstart + tagsize * 2) >> 32) Fail("ICC tag offset");
        PutTag(res, "gXYZ"); Put32(res, tagstart + tagsize); Put32(res, tagsize);
        PutTag(res, "bXYZ"); Put32(res, tagstart + tagsize * 2); Put32(res, tagsize);
      }
    }
  }
  // main content
  static const char* const kTypeStrings[8] = {"XYZ ", "desc", "text", "mluc", "para", "curv", "sf32", "gbd "};
  for (;;) {
    if (res.size() > osize || cpos > cend) Fail("ICC content out of bounds");
    if (cpos == cend) break;
    const uint8_t command = enc[cpos++];
    if (command == 1) {
      if (cpos >= cend) Fail("ICC commands out of bounds");
      const uint64_t num = in.VarInt(&cpos);
      if (num > size - pos) Fail("ICC insert out of bounds");
      res.insert(res.end(), enc.begin() + pos, enc.begin() + pos + num); pos += num;
    } else if (command == 2 || command == 3) {
      if (cpos >= cend) Fail("ICC commands out of bounds");
      const uint64_t num = in.VarInt(&cpos);
      if (num > size - pos) Fail("ICC shuffle out of bounds");
      vec<uint8_t> sh(enc.begin() + pos, enc.begin() + pos + num);
      if (num) IccShuffle(sh.data(), num, command == 2 ? 2 : 4);
      res.insert(res.end(), sh.begin(), sh.end()); pos += num;
    } else if (command == 4) {
      if (cend - cpos < 2) Fail("ICC commands out of bounds");
      const uint8_t flags = enc[cpos++];
      const size_t width = (flags & 3) + 1;
      if (width == 3) Fail("ICC predictor width");
      const int order = (flags & 12) >> 2;
      if (order == 3) Fail("ICC predictor order");
      uint64_t stride = width;
      if (flags & 16) { if (cpos >= cend) Fail("ICC commands out of bounds"); stride = in.VarInt(&cpos); if (stride < width) Fail("ICC predictor stride"); }
      if (res.empty() || ((res.size() - 1u) >> 2u) < stride) Fail("ICC predictor stride too large");
      if (cpos >= cend) Fail("ICC commands out of bounds");
      const uint64_t num = in.VarInt(&cpos);
      if (num > size - pos) Fail("ICC predict out of bounds");
      vec<uint8_t> sh(enc.begin() + pos, enc.begin() + pos + num);
      if (width > 1 && num) IccShuffle(sh.data(), num, width);
      const size_t start = res.size();
      for (size_t i = 0; i < num; i++) res.push_back((uint8_t)(IccPredict(res, start, i, stride, width, order) + sh[i]));
      pos += num;
    } else if (command == 10) {
      PutTag(res, "XYZ "); for (int i = 0; i < 4; i++) res.push_back(0);
      if (size - pos < 12) Fail("ICC XYZ out of bounds");
      res.insert(res.end(), enc.begin() + pos, enc.begin() + pos + 12); pos += 12;
    } else if (command >= 16 && command < 24) {
      PutTag(res, kTypeStrings[command - 16]); for (int i = 0; i < 4; i++) res.push_back(0);
    } else Fail("ICC command");
  }
  if (pos != size) Fail("ICC: not all data used");
  if (res.size() != osize) Fail("ICC: result size");
}

static void ReadEmbeddedIcc(Reader& r, vec<uint8_t>* icc) {
  const uint64_t enc_size = r.U64();
  if (enc_size > (1u << 28)) Fail("ICC stream too large");
  HostCode code;
  ReadEntropyCode(r, 41, &code);
  HostSymbolReader sr(r, code);
  vec<uint8_t> enc((size_t)enc_size);
  for (size_t i = 0; i < enc.size(); i++) {
    const uint32_t v = sr.Read(IccContext(i, i > 0 ? enc[i - 1] : 0, i > 1 ? enc[i - 2] : 0));
    if (v >= 256) Fail("ICC byte");
    enc[i] = (uint8_t)v;
  }
  sr.CheckFinal();
  UnpredictIcc(enc, icc);
}

void ReadTree(Reader& r, HostTree* t, size_t limit) {
  HostCode code;
  ReadEntropyCode(r, 6, &code);
  HostSymbolReader sr(r, code);
  t->nodes.clear(); t->num_leaves = 0; t->uses_wp = false; t->max_prop = 0;
  size_t pending = 1;
  while (pending > 0) {
    if (t->nodes.size() > limit) Fail("MA tree too large");
    pending--;
    int prop = (int)sr.Read(1) - 1;
    TreeNode n;
    if (prop < 0) {
      uint32_t predictor = sr.Read(2);
      if (predictor >= 14) Fail("MA tree predictor");
      int32_t offset = UnpackSigned(sr.Read(3));
      uint32_t mul_log = sr.Read(4);
      if (mul_log >= 31) Fail("MA tree mul_log");
      uint32_t mul_bits = sr.Read(5);
      if (mul_bits + 1 >= (1u << (31 - mul_log))) Fail("MA tree mul_bits");
      if (t->num_leaves >= (1u << 24)) Fail("MA tree leaves");
      n.prop = -1; n.val = offset; n.a = predictor | (t->num_leaves << 8); n.b = (mul_bits + 1u) << mul_log;
      t->num_leaves++;
      if (predictor == 6) t->uses_wp = true;
    } else {
      if (prop > 255) Fail("MA tree property");
      n.prop = prop; n.val = UnpackSigned(sr.Read(0));
      n.a = (uint32_t)(t->nodes.size() + pending + 1); n.b = (uint32_t)(t->nodes.size() + pending + 2);
      if (prop == 15) t->uses_wp = true;
      t->max_prop = std::max(t->max_prop, prop);
      pending += 2;
    }
    t->nodes.push_back(n);
  }
  sr.CheckFinal();
}

void ReadSizeHeader(Reader& r, uint32_t* xs, uint32_t* ys) {
  bool small = r.b();
  *ys = small ? (r.u(5) + 1) * 8 : r.U32({9, 1}, {13, 1}, {18, 1}, {30, 1});
  uint32_t ratio = r.u(3);
  if (ratio == 0) *xs = small ? (r.u(5) + 1) * 8 : r.U32({9, 1}, {13, 1}, {18, 1}, {30, 1});
  else {
    static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {1, 1, 10, 3, 2, 9, 4, 1};
    *xs = (uint32_t)((uint64_t)*ys * num[ratio] / den[ratio]);
  }
}
void ReadBitDepth(Reader& r, BitDepthInfo* d) {
  d->is_float = r.b();
  if (!d->is_float) { d->bits = r.U32({0, 8}, {0, 10}, {0, 12}, {6, 1}); d->exp_bits = 0; }
  else { d->bits = r.U32({0, 32}, {0, 16}, {0, 24}, {6, 1}); d->exp_bits = r.u(4) + 1; }
}
void SkipName(Reader& r) {
  uint32_t n = r.U32({0, 0}, {4, 0}, {5, 16}, {10, 48});
  for (uint32_t i = 0; i < n; i++) r.u(8);
}
void ReadCustomXY(Reader& r, double* xy) {
  for (int i = 0; i < 2; i++) {
    const uint32_t u = r.U32({19, 0}, {19, 524288}, {20, 1048576}, {21, 2097152});
    xy[i] = (double)((u & 1) ? -(int32_t)((u + 1) >> 1) : (int32_t)(u >> 1)) * 1e-6;
  }
}

void ReadTransform(Reader& r, TransformDesc* t) {
  t->id = r.u(2);
  if (t->id == 3) Fail("transform id");
  if (t->id != 2) t->begin_c = r.U32({3, 0}, {6, 8}, {10, 72}, {13, 1096});
  if (t->id == 0) { t->rct_type = r.U32({0, 6}, {2, 0}, {4, 2}, {6, 10}); if (t->rct_type >= 42) Fail("rct type"); }
  else if (t->id == 1) {
    t->num_c = r.U32({0, 1}, {0, 3}, {0, 4}, {13, 1});
    t->nb_colors = r.U32({8, 0}, {10, 256}, {12, 1280}, {16, 5376});
    t->nb_deltas = r.U32({0, 0}, {8, 1}, {10, 257}, {16, 1281});
    t->predictor = r.u(4);
    if (t->predictor >= 14) Fail("palette predictor");
  } else {
    uint32_t n = r.U32({0, 0}, {4, 1}, {6, 9}, {8, 41});
    t->squeeze.resize(n);
    for (auto& s : t->squeeze) { s.horizontal = r.b(); s.in_place = r.b(); s.begin_c = r.U32({3, 0}, {6, 8}, {10, 72}, {13, 1096}); s.num_c = r.U32({0, 1}, {0, 2}, {0, 3}, {4, 4}); }
  }
}

}  // namespace

// ---- public ------------------------------------------------------------------------------------------------------------
SigResult CheckSignature(const uint8_t* buf, size_t len) {
  // jpegxl-sys decode.rs:385 JxlSignatureCheck; utils.rs:25-33
  if (len == 0) return kSigNotEnoughBytes;
  if (buf[0] == 0xFF) {
    if (len < 2) return kSigNotEnoughBytes;
    return buf[1] == 0x0A ? kSigCodestream : kSigInvalid;
  }
  static const uint8_t sig[12] = {0, 0, 0, 0xC, 'J', 'X', 'L', ' ', 0xD, 0xA, 0x87, 0xA};
  size_t n = std::min<size_t>(len, 12);
  if (memcmp(buf, sig, n) != 0) return kSigInvalid;
  return len < 12 ? kSigNotEnoughBytes : kSigContainer;
}

// ---- free list of large codestream copies (Codestream::~Codestream) ----
namespace {
struct CsPool { std::mutex mu; std::vector<vec<uint32_t>> free; size_t bytes = 0; };
CsPool& CodestreamPool() { static CsPool* p = new CsPool; return *p; }                 // (never destroyed: decoders may outlive static destructors)
constexpr size_t kCsPoolMinWords = ((size_t)1 << 20) / 4;
size_t CsPoolMaxBytes() { static const size_t v = getenv("JXL_HIP_CODESTREAM_POOL_MB") ? (size_t)atoll(getenv("JXL_HIP_CODESTREAM_POOL_MB")) << 20 : (size_t)4 << 30; return v; }
bool PlainMalloc(const vec<uint32_t>& v) {        // the block did not come from a caller's memory manager (mm_alloc.h MmHeader)
  if (!v.data()) return false;
  const MmHeader* hd = reinterpret_cast<const MmHeader*>(reinterpret_cast<uintptr_t>(v.data()) - sizeof(MmHeader));
  return hd->free == nullptr;
}
vec<uint32_t> TakeCodestreamStorage(size_t words) {
  vec<uint32_t> out;
  if (words < kCsPoolMinWords || MmCurrent()) return out;
  CsPool& p = CodestreamPool();
  std::lock_guard<std::mutex> lock(p.mu);
  size_t best = (size_t)-1;
  for (size_t i = 0; i < p.free.size(); i++) {
    const size_t cap = p.free[i].capacity();
    if (cap >= words && cap <= words + words / 2 + 1024 && (best == (size_t)-1 || cap < p.free[best].capacity())) best = i;
  }
  if (best != (size_t)-1) { out = std::move(p.free[best]); p.bytes -= out.capacity() * 4; p.free.erase(p.free.begin() + (ptrdiff_t)best); }
  return out;
}
}  // namespace
Codestream::~Codestream() {
  if (storage.capacity() < kCsPoolMinWords || !PlainMalloc(storage)) return;
  CsPool& p = CodestreamPool();
  std::lock_guard<std::mutex> lock(p.mu);
  if (p.bytes + storage.capacity() * 4 > CsPoolMaxBytes()) return;
  p.bytes += storage.capacity() * 4;
  p.free.push_back(std::move(storage));
}

bool ExtractCodestream(const uint8_t* data, size_t size, Codestream* cs, bool* have_container, bool* has_jbrd, MetadataBoxes* boxes) {
  *have_container = false; *has_jbrd = false;
  vec<uint8_t> tmp;
  const uint8_t* src = data; size_t n = size;
  bool complete = true;
  if (!(size >= 2 && data[0] == 0xFF && data[1] == 0x0A)) {
    *have_container = true;
    size_t pos = 0;
    bool found = false, last_unbounded = false;
    while (pos + 8 <= size) {
      uint64_t bs = ((uint64_t)data[pos] << 24) | ((uint64_t)data[pos + 1] << 16) | ((uint64_t)data[pos + 2] << 8) | data[pos + 3];
      const uint8_t* type = data + pos + 4;
      size_t hdr = 8;
      if (bs == 1) {
        if (pos + 16 > size) { complete = false; break; }
        bs = 0;
        for (int i = 0; i < 8; i++) bs = (bs << 8) | data[pos + 8 + i];
        hdr = 16;
      }
      size_t end;
      if (bs == 0) { end = size; last_unbounded = true; }
      else {
        if (bs < hdr) Fail("bad box size");
        // (compared without forming pos + bs: an extended size near 2^64 must not wrap to a position before this box)
        if (bs > (uint64_t)(size - pos)) { complete = false; end = size; }
        else end = pos + (size_t)bs;
      }
      if (end < pos + hdr) { complete = false; break; }     // the box header itself is cut off: nothing of the payload is there
      if (!memcmp(type, "jxlc", 4)) { tmp.insert(tmp.end(), data + pos + hdr, data + end); found = true; }
      else if (!memcmp(type, "jxlp", 4)) { if (end >= pos + hdr + 4) { tmp.insert(tmp.end(), data + pos + hdr + 4, data + end); found = true; } }
      else if (!memcmp(type, "jbrd", 4)) { *has_jbrd = true; if (boxes) boxes->jbrd.assign(data + pos + hdr, data + end); }
      else if (boxes) {
        // metadata a reconstructed JPEG gets its APP1 markers from (decode.cc box handling: the first box of a kind counts)
        const bool brob = !memcmp(type, "brob", 4) && end >= pos + hdr + 4;
        const uint8_t* real = brob ? data + pos + hdr : type;
        const size_t body = pos + hdr + (brob ? 4 : 0);
        if (!memcmp(real, "Exif", 4) && !boxes->have_exif) { boxes->have_exif = true; boxes->exif_brob = brob; boxes->exif.assign(data + body, data + end); }
        else if (!memcmp(real, "xml ", 4) && !boxes->have_xml) { boxes->have_xml = true; boxes->xml_brob = brob; boxes->xml.assign(data + body, data + end); }
      }
      pos = end;
    }
    (void)last_unbounded;
    if (!found) complete = false;
    src = tmp.data(); n = tmp.size();
  }
  cs->size = n;
  const size_t words = (n + 3) / 4 + 20;       // 80 zero bytes after the stream: the HF kernel's bit-stream ring prefetches up to 64 bytes ahead
  cs->storage = TakeCodestreamStorage(words);
  if (cs->storage.capacity() >= words) cs->storage.resize(words);      // (a recycled block: only what lies beyond its old size is zero-filled)
  else cs->storage.assign(words, 0);
  if (n) memcpy(cs->data(), src, n);
  memset(cs->data() + n, 0, words * 4 - n);
  return complete;
}

void ParseImageHeader(const Codestream& cs, ImageHeader* ih, uint64_t* frame_bitpos) {
  Reader r(cs, 0);
  if (r.u(16) != 0x0AFF) Fail("not a JPEG XL codestream");
  ReadSizeHeader(r, &ih->xsize, &ih->ysize);
  bool all_default = r.b();
  bool extra_fields = false;
  if (!all_default) {
    extra_fields = r.b();
    if (extra_fields) {
      ih->orientation = r.u(3) + 1;
      if (r.b()) ReadSizeHeader(r, &ih->intrinsic_x, &ih->intrinsic_y);
      ih->have_preview = r.b();
      if (ih->have_preview) {   // headers.cc PreviewHeader
        const bool div8 = r.b();
        auto dim = [&]() { return div8 ? 8 * r.U32({0, 16}, {0, 32}, {5, 1}, {9, 33}) : r.U32({6, 1}, {8, 65}, {10, 321}, {12, 1345}); };
        ih->preview_y = dim();
        const uint32_t ratio = r.u(3);
        if (ratio == 0) ih->preview_x = dim();
        else {
          static const uint32_t num[8] = {0, 1, 12, 4, 3, 16, 5, 2}, den[8] = {1, 1, 10, 3, 2, 9, 4, 1};
          ih->preview_x = (uint32_t)((uint64_t)ih->preview_y * num[ratio] / den[ratio]);
        }
        if (ih->preview_x > 4096 || ih->preview_y > 4096) Fail("preview too large");
      }
      ih->have_animation = r.b();
      if (ih->have_animation) {
        ih->tps_num = r.U32({0, 100}, {0, 1000}, {10, 1}, {30, 1});
        ih->tps_den = r.U32({0, 1}, {0, 1001}, {8, 1}, {10, 1});
        ih->num_loops = r.U32({0, 0}, {3, 0}, {16, 0}, {32, 0});
        ih->have_timecodes = r.b();
      }
    }
    ReadBitDepth(r, &ih->depth);
    r.b();  // modular_16bit_buffers
    uint32_t nextra = r.U32({0, 0}, {0, 1}, {4, 2}, {12, 1});
    ih->extra.resize(nextra);
    for (auto& e : ih->extra) {
      if (r.b()) continue;  // default 8-bit alpha
      e.type = r.Enum();
      ReadBitDepth(r, &e.depth);
      e.dim_shift = r.U32({0, 0}, {0, 3}, {0, 4}, {3, 1});
      { const uint32_t n = r.U32({0, 0}, {4, 0}, {5, 16}, {10, 48}); for (uint32_t i = 0; i < n; i++) e.name.push_back((char)r.u(8)); }     // (JxlDecoderGetExtraChannelName)
      if (e.type == 0) e.alpha_associated = r.b();
      if (e.type == 2) for (int i = 0; i < 4; i++) e.spot[i] = r.F16();
      if (e.type == 5) e.cfa_channel = r.U32({0, 1}, {2, 0}, {4, 3}, {8, 19});
    }
    ih->xyb_encoded = r.b();
    ih->color_default = r.b();
    if (!ih->color_default) {
      ih->want_icc = r.b();
      ih->color_space = r.Enum();
      if (!ih->want_icc) {
        if (ih->color_space != 2) { ih->white_point = r.Enum(); if (ih->white_point == 2) ReadCustomXY(r, ih->white_xy); }
        if (ih->color_space != 2 && ih->color_space != 1) { ih->primaries = r.Enum(); if (ih->primaries == 2) for (int i = 0; i < 3; i++) ReadCustomXY(r, ih->prim_xy + 2 * i); }
        if (ih->color_space != 2) { ih->have_gamma = r.b(); if (ih->have_gamma) ih->gamma = r.u(24); else ih->tf = r.Enum(); }
        ih->rendering_intent = r.Enum();
      }
    }
    if (extra_fields && !r.b()) {
      ih->intensity_target = r.F16(); ih->min_nits = r.F16(); ih->relative_to_max_display = r.b(); ih->linear_below = r.F16();
      if (!(ih->intensity_target > 0.0f)) Fail("intensity target must be positive");
    }
    r.SkipExtensions();
  }
  static const float kInv[9] = {11.031566901960783f,  -9.866943921568629f, -0.16462299647058826f, -3.254147380392157f, 4.418770392156863f,
                                -0.16462299647058826f, -3.6588512862745097f, 2.7129230470588235f, 1.9459282392156863f};
  for (int i = 0; i < 9; i++) ih->opsin_inv[i] = kInv[i];
  for (int i = 0; i < 3; i++) ih->opsin_bias[i] = -0.0037930732552754493f;
  ih->quant_bias[0] = 1.0f - 0.05465007330715401f; ih->quant_bias[1] = 1.0f - 0.07005449891748593f;
  ih->quant_bias[2] = 1.0f - 0.049935103337343655f; ih->quant_bias[3] = 0.145f;
  if (!r.b()) {  // default_m == false
    if (ih->xyb_encoded && !r.b()) {
      for (int i = 0; i < 9; i++) ih->opsin_inv[i] = r.F16();
      for (int i = 0; i < 3; i++) ih->opsin_bias[i] = r.F16();
      for (int i = 0; i < 4; i++) ih->quant_bias[i] = r.F16();
    }
    uint32_t cw = r.u(3);
    static const int kCount[3] = {15, 55, 210};
    for (int k = 0; k < 3; k++) if (cw >> k & 1) { ih->up_weights[k].resize(kCount[k]); for (float& v : ih->up_weights[k]) v = r.F16(); }
  }
  if (ih->want_icc) ReadEmbeddedIcc(r, &ih->icc);
  r.align();
  *frame_bitpos = r.pos();
}

static void ParseLfGlobal(Reader& r, const ImageHeader& ih, FramePlan* p);
static void ParseLocalModularStreams(const Codestream& cs, FramePlan* p);

void ParseFrameStart(const Codestream& cs, const ImageHeader& ih, uint64_t frame_bitpos, FramePlan* p, bool header_and_toc_only, bool allow_partial) {
  const bool skip = header_and_toc_only;
  Reader r(cs, frame_bitpos);
  const size_t num_extra = ih.extra.size();
  const bool xyb = ih.xyb_encoded;
  uint32_t fx = ih.xsize, fy = ih.ysize;
  vec<uint32_t> ec_ups(num_extra, 1);
  bool all_default = r.b();
  if (!xyb) { p->x_qm_scale = 2; p->b_qm_scale = 2; }
  if (!all_default) {
    p->frame_type = r.u(2);
    p->modular = r.u(1) != 0;
    p->flags = r.U64();
    if (!xyb) p->do_ycbcr = r.b();
    const bool use_lf_frame = (p->flags & 32) != 0;
    if (p->do_ycbcr && !use_lf_frame) for (int i = 0; i < 3; i++) p->jpeg_upsampling[i] = r.u(2);
    if (!use_lf_frame) {
      p->upsampling = r.U32({0, 1}, {0, 2}, {0, 4}, {0, 8});
      for (auto& e : ec_ups) e = r.U32({0, 1}, {0, 2}, {0, 4}, {0, 8});
    }
    if (p->modular) p->group_size_shift = r.u(2);
    if (!p->modular && xyb) { p->x_qm_scale = r.u(3); p->b_qm_scale = r.u(3); }
    if (p->frame_type != 2) {
      p->num_passes = r.U32({0, 1}, {0, 2}, {0, 3}, {3, 4});
      if (p->num_passes != 1) {
        p->num_ds = r.U32({0, 0}, {0, 1}, {0, 2}, {1, 3});
        for (uint32_t i = 0; i + 1 < p->num_passes; i++) p->pass_shift[i] = r.u(2);
        for (uint32_t i = 0; i < p->num_ds; i++) p->downsample[i] = r.U32({0, 1}, {0, 2}, {0, 4}, {0, 8});
        for (uint32_t i = 0; i < p->num_ds; i++) p->ds_last_pass[i] = r.U32({0, 0}, {0, 1}, {0, 2}, {3, 0});
      }
    }
    if (p->num_passes > 11) Fail("number of passes");
    for (uint32_t pass = 0; pass < p->num_passes; pass++) {     // passes.h GetDownsamplingBracket
      int32_t mins = 3, maxs = 2;
      for (uint32_t i = 0;; i++) {
        for (uint32_t j = 0; j < p->num_ds; j++) if (i == p->ds_last_pass[j]) mins = p->downsample[j] == 8 ? 3 : p->downsample[j] == 4 ? 2 : p->downsample[j] == 2 ? 1 : 0;
        if (i + 1 == p->num_passes) mins = 0;
        if (i == pass) break;
        maxs = mins - 1;
      }
      p->pass_min_shift[pass] = mins; p->pass_max_shift[pass] = maxs;
    }
    p->mod_pass = p->num_passes - 1;
    for (uint32_t pass = 0; pass < p->num_passes; pass++) if (p->pass_min_shift[pass] <= 0 && p->pass_max_shift[pass] >= 0) { p->mod_pass = pass; break; }
    bool partial = false;
    p->use_lf_frame = use_lf_frame;
    if (p->frame_type == 1) {
      p->lf_level = r.U32({0, 1}, {0, 2}, {0, 3}, {0, 4});
      const uint32_t d = 1u << (3 * p->lf_level);      // (frame_header.cc ToFrameDimensions: the image size divided by 8^level, rounded up)
      fx = (fx + d - 1) / d; fy = (fy + d - 1) / d;
    } else if (r.b()) {
      p->have_crop = true;
      if (p->frame_type != 2) {
        p->x0 = UnpackSigned(r.U32({8, 0}, {11, 256}, {14, 2304}, {30, 18688}));
        p->y0 = UnpackSigned(r.U32({8, 0}, {11, 256}, {14, 2304}, {30, 18688}));
      }
      fx = r.U32({8, 0}, {11, 256}, {14, 2304}, {30, 18688});
      fy = r.U32({8, 0}, {11, 256}, {14, 2304}, {30, 18688});
      if (fx == 0 || fy == 0) Fail("empty frame");
      partial = p->x0 > 0 || p->y0 > 0 || (int64_t)fx + p->x0 < (int64_t)ih.xsize || (int64_t)fy + p->y0 < (int64_t)ih.ysize;
    }
    p->ec_blend.assign(num_extra, BlendInfoH());
    if (p->frame_type == 0 || p->frame_type == 3) {
      for (size_t i = 0; i < 1 + num_extra; i++) {
        BlendInfoH& b = i == 0 ? p->blend : p->ec_blend[i - 1];
        b.mode = r.U32({0, 0}, {0, 1}, {0, 2}, {2, 3});
        if (b.mode > 4) Fail("blend mode");
        if (num_extra > 0 && (b.mode == 2 || b.mode == 3)) b.alpha_channel = r.U32({0, 0}, {0, 1}, {0, 2}, {3, 3});
        if (num_extra > 0 && (b.mode == 2 || b.mode == 3 || b.mode == 4)) b.clamp = r.b();
        if (b.mode != 0 || partial) b.source = r.u(2);
        if (b.alpha_channel >= std::max<size_t>(num_extra, 1)) Fail("blend alpha channel");
      }
      if (ih.have_animation) { p->duration = r.U32({0, 0}, {0, 1}, {8, 0}, {32, 0}); if (ih.have_timecodes) p->timecode = r.u(32); }
      p->is_last = r.b();
    } else p->is_last = false;
    if (p->frame_type != 1 && !p->is_last) p->save_as_reference = r.u(2);
    bool can_ref = !p->is_last && p->frame_type != 1 && (p->duration == 0 || p->save_as_reference != 0);
    bool full_replace = (p->frame_type == 0 || p->frame_type == 3) && p->blend.mode == 0 && !partial;
    if (p->frame_type == 2 || (can_ref && full_replace)) p->save_before_ct = r.b();
    {  // frame name (JxlDecoderGetFrameName)
      const uint32_t n = r.U32({0, 0}, {4, 0}, {5, 16}, {10, 48});
      p->name.clear();
      for (uint32_t i = 0; i < n; i++) p->name.push_back((char)r.u(8));
    }
    if (!r.b()) {  // RestorationFilter not all_default
      LoopFilterParams& lf = p->lf;
      lf.gab = r.b();
      if (lf.gab && r.b()) for (int i = 0; i < 6; i++) lf.gab_w[i] = r.F16();
      lf.epf_iters = r.u(2);
      if (lf.epf_iters > 0) {
        if (!p->modular && r.b()) for (int i = 0; i < 8; i++) lf.sharp_lut[i] = r.F16();
        if (r.b()) { for (int i = 0; i < 3; i++) lf.channel_scale[i] = r.F16(); r.F16(); r.F16(); }
        if (r.b()) { if (!p->modular) lf.quant_mul = r.F16(); lf.pass0_sigma_scale = r.F16(); lf.pass2_sigma_scale = r.F16(); lf.border_sad_mul = r.F16(); }
        if (p->modular) p->sigma_for_modular = r.F16();
      }
      r.SkipExtensions();
    }
    r.SkipExtensions();
    if (use_lf_frame && p->modular) Fail("use_lf_frame on a Modular frame");
    if (use_lf_frame && p->frame_type == 1 && p->lf_level >= 4) Fail("LF frame level");
    if (p->frame_type == 1 && (p->upsampling != 1 || (p->flags & (1 | 2 | 16)))) Fail("LF frame with upsampling / image features");
    if (!skip) for (auto e : ec_ups) if (e != p->upsampling) Unsupported("extra-channel upsampling different from the colour upsampling");
  }
  p->frame_w = fx; p->frame_h = fy;
  if (p->upsampling != 1) { fx = (fx + p->upsampling - 1) / p->upsampling; fy = (fy + p->upsampling - 1) / p->upsampling; }   // coded size
  if (!skip && p->modular && (p->lf.gab || p->lf.epf_iters)) Unsupported("restoration filters on a Modular frame");
  if (p->num_passes > 11) Fail("number of passes");
  p->width = fx; p->height = fy;
  p->group_dim = p->modular ? (128u << p->group_size_shift) : 256u;
  p->xgroups = (fx + p->group_dim - 1) / p->group_dim; p->ygroups = (fy + p->group_dim - 1) / p->group_dim;
  p->num_groups = p->xgroups * p->ygroups;
  p->xlfgroups = (fx + p->group_dim * 8 - 1) / (p->group_dim * 8); p->ylfgroups = (fy + p->group_dim * 8 - 1) / (p->group_dim * 8);
  p->num_lf_groups = p->xlfgroups * p->ylfgroups;
  p->bw = (fx + 7) / 8; p->bh = (fy + 7) / 8;
  if (p->do_ycbcr && !p->modular) {
    static const uint32_t kH[4] = {0, 1, 1, 0}, kV[4] = {0, 1, 0, 1};
    uint32_t maxhs = 0, maxvs = 0;
    for (int c = 0; c < 3; c++) { maxhs = std::max(maxhs, kH[p->jpeg_upsampling[c]]); maxvs = std::max(maxvs, kV[p->jpeg_upsampling[c]]); }
    for (int c = 0; c < 3; c++) { p->hs[c] = maxhs - kH[p->jpeg_upsampling[c]]; p->vs[c] = maxvs - kV[p->jpeg_upsampling[c]]; p->subsampled |= p->hs[c] || p->vs[c]; }
    if (p->subsampled) {
      p->bw = ((fx + (8u << maxhs) - 1) / (8u << maxhs)) << maxhs; p->bh = ((fy + (8u << maxvs) - 1) / (8u << maxvs)) << maxvs;
      if (!skip && (!(p->flags & 128) || p->lf.gab || p->lf.epf_iters || p->upsampling != 1 || p->num_passes != 1))
        Unsupported("chroma subsampling together with LF smoothing / restoration filters / upsampling / progressive passes");
    }
  } else if (p->do_ycbcr && !skip) for (int c = 0; c < 3; c++) if (p->jpeg_upsampling[c]) Unsupported("chroma subsampling in a Modular frame");
  // ---- TOC
  p->single_section = p->num_groups == 1 && p->num_passes == 1;
  size_t n = p->single_section ? 1 : 1 + p->num_lf_groups + 1 + (size_t)p->num_groups * p->num_passes;
  // (every TOC entry takes at least 12 bits: a header that announces more sections than the rest of the stream could list is cut off or damaged — said before
  // anything of that size is allocated)
  if (n > ((uint64_t)cs.size * 8 - std::min<uint64_t>(r.pos(), (uint64_t)cs.size * 8)) / 12 + 1) throw ParseError("truncated", false);
  // toc.cc ReadToc: an optional Lehmer-coded permutation says where logical section i is stored
  vec<uint32_t> perm;
  if (r.b()) {
    HostCode pc;
    ReadEntropyCode(r, 8, &pc);
    HostSymbolReader sr(r, pc);
    auto ctxof = [](uint32_t v) { uint32_t t = v == 0 ? 0 : 1 + FloorLog2(v); return std::min<uint32_t>(t, 7); };
    const uint32_t end = sr.Read(ctxof((uint32_t)n));
    if (end > n) Fail("TOC permutation size");
    vec<uint32_t> lehmer(n, 0), temp(n);
    uint32_t last = 0;
    for (size_t i = 0; i < end; i++) { lehmer[i] = sr.Read(ctxof(last)); last = lehmer[i]; if (lehmer[i] >= n - i) Fail("TOC lehmer code"); }
    sr.CheckFinal();
    // perm[i] = the lehmer[i]-th element still unused, in O(n log n): a Fenwick tree over "unused" flags, descended from the top bit (erasing from a vector is quadratic —
    // a damaged header that claims millions of groups made the parser spin for an hour; found by tests/fuzz/fuzz_host.cc)
    size_t top = 1;
    while (top * 2 <= n) top *= 2;
    for (size_t i = 0; i < n; i++) temp[i] = (uint32_t)((i + 1) & (~(i + 1) + 1));      // tree of all ones: node i (1-based) covers lowbit(i) elements
    perm.resize(n);
    for (size_t i = 0; i < n; i++) {
      size_t pos = 0, rem = lehmer[i];                      // the element with exactly `rem` unused elements in front of it
      for (size_t step = top; step; step >>= 1)
        if (pos + step <= n && temp[pos + step - 1] <= rem) { pos += step; rem -= temp[pos - 1]; }
      perm[i] = (uint32_t)pos;                               // 0-based index of that element
      for (size_t k = pos + 1; k <= n; k += k & (~k + 1)) temp[k - 1]--;
    }
  }
  r.align();
  vec<uint64_t> sizes(n);
  for (auto& s : sizes) s = r.U32({10, 0}, {14, 1024}, {22, 17408}, {30, 4211712});
  r.align();
  uint64_t off = r.pos() / 8;
  vec<Section> phys(n);
  for (size_t i = 0; i < n; i++) { phys[i] = {off, sizes[i]}; off += sizes[i]; }
  p->sections.resize(n);
  for (size_t i = 0; i < n; i++) p->sections[i] = perm.empty() ? phys[i] : phys[perm[i]];
  if (off > cs.size) {
    // the stream ends inside this frame.  A VarDCT frame whose LfGlobal, LfGroup and HfGlobal sections are all there can be shown without its AC groups (the kDC step of
    // progressive decoding, JxlDecoderFlushImage): only for the plain case — sections in file order, no Modular extra channels riding in the PassGroups, no LF frame
    const size_t lf_last = 1 + p->num_lf_groups;
    const bool lf_complete = !p->single_section && !p->modular && perm.empty() && n > lf_last && phys[lf_last].offset + phys[lf_last].size <= cs.size;
    if (!(allow_partial && !skip && lf_complete && ih.extra.empty() && !p->use_lf_frame)) throw ParseError("truncated", false);
    p->partial = true;
    for (size_t i = lf_last + 1; i < n; i++) p->partial_ac_sections += phys[i].offset + phys[i].size <= cs.size;
  }
  p->frame_end_bitpos = off * 8;
  if (skip) return;
  // ---- LfGlobal
  Reader rg(cs, p->sections[0].offset * 8);
  rg.limit_bits = (p->sections[0].offset + p->sections[0].size) * 8;
  ParseLfGlobal(rg, ih, p);
  p->end_bitpos = rg.pos();
  ParseLocalModularStreams(cs, p);
  if (p->max_prop >= 16 + 4 * kMaxModRefs) Unsupported("MA tree property beyond the supported previous-channel references");
  if (!p->single_section && !p->modular) ParseHfGlobal(cs, ih, p->sections[1 + p->num_lf_groups].offset * 8, p);
}

// dec_patch_dictionary.cc PatchDictionary::Decode — contexts {0 #refs, 1 reference frame, 2 size - 1, 3 position in the reference,
// 4 first position, 5 blend mode, 6 position delta, 7 count - 1, 8 alpha channel, 9 clamp}
static void ParsePatches(Reader& r, size_t num_extra, size_t frame_pixels, FrameFeatures* f) {
  HostCode code;
  ReadEntropyCode(r, 10, &code);
  HostSymbolReader sr(r, code);
  const uint32_t num = sr.Read(0);
  if ((uint64_t)num > frame_pixels + 1024) Fail("too many patches");
  f->patches.resize(num);
  size_t total = 0;
  for (PatchRefH& pr : f->patches) {
    pr.ref = sr.Read(1);
    if (pr.ref >= 4) Fail("patch reference frame");
    pr.x0 = sr.Read(3); pr.y0 = sr.Read(3);
    // (sizes in 64 bits: a 0xFFFFFFFF token must not wrap to an empty patch — dec_patch_dictionary.cc computes in size_t and rejects
    // patches larger than the reference frame; no frame is larger than 2^30 in either direction)
    const uint64_t xs = (uint64_t)sr.Read(2) + 1, ys = (uint64_t)sr.Read(2) + 1;
    if (xs > (1u << 30) || ys > (1u << 30) || (uint64_t)pr.x0 + xs > (1ull << 31) || (uint64_t)pr.y0 + ys > (1ull << 31)) Fail("patch size");
    pr.xsize = (uint32_t)xs; pr.ysize = (uint32_t)ys;
    const uint32_t count = sr.Read(7) + 1;
    total += count;
    if (total > frame_pixels + 1024) Fail("too many patch positions");
    pr.pos.resize(count);
    for (uint32_t i = 0; i < count; i++) {
      PatchPosH& pp = pr.pos[i];
      if (i == 0) { pp.x = sr.Read(4); pp.y = sr.Read(4); }
      else { pp.x = pr.pos[i - 1].x + UnpackSigned(sr.Read(6)); pp.y = pr.pos[i - 1].y + UnpackSigned(sr.Read(6)); }
      if (pp.x < 0 || pp.y < 0) Fail("patch position");
      pp.blend.resize(1 + num_extra);
      for (PatchBlendH& b : pp.blend) {
        b.mode = sr.Read(5);
        if (b.mode >= 8) Fail("patch blend mode");
        const bool uses_alpha = b.mode >= 4;
        if (uses_alpha && num_extra > 1) { b.alpha_channel = sr.Read(8); if (b.alpha_channel >= num_extra) Fail("patch alpha channel"); }
        if (uses_alpha || b.mode == 3) b.clamp = sr.Read(9) != 0;
        if (uses_alpha && num_extra == 0) Fail("alpha patch blending without extra channels");
      }
    }
  }
  sr.CheckFinal();
}

// splines.cc Splines::Decode — contexts {0 quantisation adjustment, 1 starting position, 2 #splines - 1, 3 #control points,
// 4 control point double deltas, 5 DCT coefficients}
static void ParseSplines(Reader& r, size_t num_pixels, FrameFeatures* f) {
  HostCode code;
  ReadEntropyCode(r, 6, &code);
  HostSymbolReader sr(r, code);
  const size_t num = 1 + (size_t)sr.Read(2);
  const size_t max_cp = std::min<size_t>(1u << 20, num_pixels / 2);
  if (num > max_cp) Fail("too many splines");
  f->spline_start.resize(num);
  int64_t lx = 0, ly = 0;
  for (size_t i = 0; i < num; i++) {
    int64_t x, y;
    if (i == 0) { x = sr.Read(1); y = sr.Read(1); }
    else { x = lx + UnpackSigned(sr.Read(1)); y = ly + UnpackSigned(sr.Read(1)); }
    if (std::llabs(x) >= (1 << 23) || std::llabs(y) >= (1 << 23)) Fail("spline starting point");
    f->spline_start[i] = {x, y};
    lx = x; ly = y;
  }
  f->spline_quant_adjust = UnpackSigned(sr.Read(0));
  f->splines.resize(num);
  size_t total = 0;
  for (SplineH& q : f->splines) {
    const size_t n = sr.Read(3);
    total += n;
    if (total > max_cp) Fail("too many spline control points");
    q.control_points.resize(n);
    for (auto& cp : q.control_points) {
      cp.first = UnpackSigned(sr.Read(4)); cp.second = UnpackSigned(sr.Read(4));
      if (std::llabs(cp.first) >= (1 << 30) || std::llabs(cp.second) >= (1 << 30)) Fail("spline delta");
    }
    for (int c = 0; c < 3; c++) for (int i = 0; i < 32; i++) q.color_dct[c][i] = UnpackSigned(sr.Read(5));
    for (int i = 0; i < 32; i++) q.sigma_dct[i] = UnpackSigned(sr.Read(5));
  }
  sr.CheckFinal();
}

static void ParseLfGlobal(Reader& r, const ImageHeader& ih, FramePlan* p) {
  // dec_frame.cc ProcessDCGlobal: image features first — patches, splines, noise parameters
  if (p->flags & 2) ParsePatches(r, ih.extra.size(), (size_t)p->width * p->height, &p->feat);
  if (p->flags & 16) ParseSplines(r, (size_t)p->width * p->height, &p->feat);
  if (p->flags & 1) { p->feat.has_noise = true; for (float& v : p->feat.noise_lut) v = (float)r.u(10) * (1.0f / 1024.0f); }
  if (!r.b()) for (int c = 0; c < 3; c++) p->m_lf[c] = r.F16() * (1.0f / 128.0f);
  BlockCtxDev& b = p->bcm;
  memset(&b, 0, sizeof(b));
  if (!p->modular) {
    p->global_scale = r.U32({11, 1}, {11, 2049}, {12, 4097}, {16, 8193});
    p->quant_lf = r.U32({0, 16}, {5, 1}, {8, 1}, {16, 1});
    static const uint8_t kDefault[39] = {0, 1, 2, 2, 3, 3, 4, 5, 6, 6, 6, 6, 6, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14, 7, 8, 9, 9, 10, 11, 12, 13, 14, 14, 14, 14, 14};
    b.num_lf_ctxs = 1; b.num_ctxs = 15;
    memcpy(b.ctx_map, kDefault, 39);
    if (!r.b()) {
      for (int j = 0; j < 3; j++) {
        b.n_lf_thr[j] = r.u(4);
        for (uint32_t i = 0; i < b.n_lf_thr[j]; i++) b.lf_thr[j][i] = UnpackSigned(r.U32({4, 0}, {8, 16}, {16, 272}, {32, 65808}));
      }
      b.n_qf_thr = r.u(4);
      for (uint32_t i = 0; i < b.n_qf_thr; i++) b.qf_thr[i] = r.U32({2, 0}, {3, 4}, {5, 12}, {8, 44}) + 1;
      b.num_lf_ctxs = (b.n_lf_thr[0] + 1) * (b.n_lf_thr[1] + 1) * (b.n_lf_thr[2] + 1);
      size_t n = (size_t)39 * b.num_lf_ctxs * (b.n_qf_thr + 1);
      if (n > sizeof(b.ctx_map)) Fail("block context map too large");
      vec<uint8_t> map; uint32_t nc = 0;
      ReadContextMap(r, (uint32_t)n, map, &nc);
      if (nc > 16) Fail("too many block contexts");
      memcpy(b.ctx_map, map.data(), n);
      b.num_ctxs = nc;
    }
    if (!r.b()) {
      p->color_factor = r.U32({0, 84}, {0, 256}, {8, 2}, {16, 258});
      p->base_x = r.F16(); p->base_b = r.F16();
      p->ytox_lf = (int32_t)r.u(8) - 128; p->ytob_lf = (int32_t)r.u(8) - 128;
    }
  }
  // GlobalModular (dec_modular.cc DecodeGlobalInfo)
  p->has_global_tree = r.b();
  uint32_t nb_color = 0;
  if (p->modular) nb_color = (ih.color_space == 1 && !ih.xyb_encoded && !p->do_ycbcr) ? 1 : 3;
  p->nb_color_channels = nb_color;
  if (p->has_global_tree) {
    size_t limit = std::min<size_t>(1u << 22, 1024 + (size_t)p->width * p->height * (nb_color + ih.extra.size()) / 16);
    ReadTree(r, &p->tree, limit);
    ReadEntropyCode(r, p->tree.num_leaves, &p->tree_code);
    p->max_prop = p->tree.max_prop;
  }
  p->local_streams.clear();
  p->gchannels.clear();
  for (uint32_t c = 0; c < nb_color + ih.extra.size(); c++) p->gchannels.push_back({p->width, p->height, 0, 0});
  p->nb_meta_channels = 0;
  p->global_decodable = 0;
  p->gwp = WPHeader{16, 10, {7, 7, 7, 0, 0}, {13, 12, 12, 12}};
  if (p->gchannels.empty()) { p->global_data_bitpos = r.pos(); return; }
  // GroupHeader
  p->g_use_global_tree = r.b();
  if (!r.b()) {
    p->gwp.p1 = r.u(5); p->gwp.p2 = r.u(5);
    for (int i = 0; i < 5; i++) p->gwp.p3[i] = r.u(5);
    for (int i = 0; i < 4; i++) p->gwp.w[i] = r.u(4);
  }
  uint32_t nt = r.U32({0, 0}, {0, 1}, {4, 2}, {8, 18});
  p->gtransforms.resize(nt);
  for (auto& t : p->gtransforms) {
    ReadTransform(r, &t);
    // apply to the channel list (transform.cc MetaApply)
    auto& ch = p->gchannels;
    if (t.id == 0) { if (t.begin_c + 3 > ch.size()) Fail("rct range"); }
    else if (t.id == 1) {
      uint32_t endc = t.begin_c + t.num_c - 1;
      if (endc >= ch.size()) Fail("palette range");
      if (t.begin_c < p->nb_meta_channels) Unsupported("palette over meta channels");
      p->nb_meta_channels += 1;
      ch.erase(ch.begin() + t.begin_c + 1, ch.begin() + endc + 1);
      ch.insert(ch.begin(), FramePlan::ModChannel{t.nb_colors, t.num_c, -1, 0});
    } else {
      // Squeeze (squeeze.cc MetaSqueeze): zero explicit steps = the default chain derived from the channel sizes
      if (t.squeeze.empty()) {
        const uint32_t first = p->nb_meta_channels;
        if (first >= ch.size()) Fail("squeeze without channels");
        const uint32_t nb = (uint32_t)ch.size() - first;
        uint32_t w = ch[first].w, h = ch[first].h;
        if (nb > 2 && ch[first + 1].w == w && ch[first + 1].h == h) { t.squeeze.push_back({1, 0, first + 1, 2}); t.squeeze.push_back({0, 0, first + 1, 2}); }
        if (w <= h && h > 8) { t.squeeze.push_back({0, 1, first, nb}); h = (h + 1) / 2; }
        while (w > 8 || h > 8) {
          if (w > 8) { t.squeeze.push_back({1, 1, first, nb}); w = (w + 1) / 2; }
          if (h > 8) { t.squeeze.push_back({0, 1, first, nb}); h = (h + 1) / 2; }
        }
      }
      for (const SqueezeStep& q : t.squeeze) {
        const uint32_t endc = q.begin_c + q.num_c - 1;
        if (endc >= ch.size()) Fail("squeeze range");
        if (q.begin_c < p->nb_meta_channels) Unsupported("squeeze over meta channels");
        const uint32_t offset = q.in_place ? endc + 1 : (uint32_t)ch.size();
        for (uint32_t c = q.begin_c; c <= endc; c++) {
          FramePlan::ModChannel& a = ch[c];
          if (a.hshift > 30 || a.vshift > 30) Fail("squeeze depth");
          FramePlan::ModChannel res = a;
          if (q.horizontal) { res.w = a.w - (a.w + 1) / 2; a.w = (a.w + 1) / 2; a.hshift++; res.hshift = a.hshift; }
          else { res.h = a.h - (a.h + 1) / 2; a.h = (a.h + 1) / 2; a.vshift++; res.vshift = a.vshift; }
          ch.insert(ch.begin() + offset + (c - q.begin_c), res);
        }
        if (ch.size() > 4096) Fail("too many squeezed channels");
      }
    }
  }
  if (!p->g_use_global_tree) {
    // the global stream brings its own tree and code (encoding.cc ModularDecode)
    FramePlan::LocalStream ls;
    size_t pixels = 0;
    for (auto& c : p->gchannels) pixels += (size_t)c.w * c.h;
    ReadTree(r, &ls.tree, std::min<size_t>(1u << 22, 1024 + pixels));
    ReadEntropyCode(r, ls.tree.num_leaves, &ls.code);
    ls.unit = 0;
    p->max_prop = std::max(p->max_prop, ls.tree.max_prop);
    p->local_streams.push_back(std::move(ls));
  } else if (!p->has_global_tree) Fail("global tree missing");
  uint32_t end = (uint32_t)p->gchannels.size();
  for (uint32_t i = 0; i < p->gchannels.size(); i++) {
    const auto& c = p->gchannels[i];
    if (i >= p->nb_meta_channels && (c.w > p->group_dim || c.h > p->group_dim)) { end = i; break; }
  }
  p->global_decodable = end;
  p->global_data_bitpos = r.pos();
  if (!p->local_streams.empty()) p->local_streams[0].data_bitpos = r.pos();
}

// Sections of a Modular frame start with their Modular sub-stream: read the GroupHeader of every LfGroup / PassGroup unit that
// holds channels and, where it says use_global_tree = 0, the stream's own tree and code (dec_modular.cc DecodeGroup,
// encoding.cc ModularDecode).  VarDCT frames keep such streams behind data only the device decodes: not parsed here.
static void ParseLocalModularStreams(const Codestream& cs, FramePlan* p) {
  if (!p->modular || p->single_section) return;
  p->mod_units_scanned = true;
  if (p->global_decodable >= p->gchannels.size()) return;
  const uint32_t total = p->NumModUnits();
  for (uint32_t unit = 0; unit < total; unit++) {
    const bool is_lf = unit < p->num_lf_groups;
    const uint32_t pass = is_lf ? 0 : (unit - p->num_lf_groups) / p->num_groups;
    const uint32_t g = is_lf ? unit : (unit - p->num_lf_groups) % p->num_groups;
    const uint32_t dim = is_lf ? p->group_dim * 8 : p->group_dim, cols = is_lf ? p->xlfgroups : p->xgroups;
    const uint32_t x0 = (g % cols) * dim, y0 = (g / cols) * dim;
    const int min_shift = is_lf ? 3 : p->pass_min_shift[pass], max_shift = is_lf ? 1000 : p->pass_max_shift[pass];
    size_t pixels = 0;
    for (size_t c = p->global_decodable; c < p->gchannels.size(); c++) {
      const auto& m = p->gchannels[c];
      if (m.w == 0 || m.h == 0) continue;
      const int shift = std::min(m.hshift, m.vshift);
      if (shift < min_shift || shift > max_shift) continue;
      const uint32_t rx = x0 >> m.hshift, ry = y0 >> m.vshift;
      if (rx >= m.w || ry >= m.h) continue;
      pixels += (size_t)std::min(dim >> m.hshift, m.w - rx) * std::min(dim >> m.vshift, m.h - ry);
    }
    if (pixels == 0) continue;
    const Section& sec = p->sections[is_lf ? 1 + g : 2 + p->num_lf_groups + pass * p->num_groups + g];
    Reader r(cs, sec.offset * 8);
    r.limit_bits = (sec.offset + sec.size) * 8;
    const bool global_tree = r.b();
    if (global_tree && !p->has_global_tree) Fail("global tree missing");
    if (!r.b()) { r.u(5); r.u(5); for (int i = 0; i < 5; i++) r.u(5); for (int i = 0; i < 4; i++) r.u(4); }   // WPHeader (read again on the device)
    const uint32_t nt = r.U32({0, 0}, {0, 1}, {4, 2}, {8, 18});
    if (nt) p->mod_local_transforms = true;          // (the unit decodes into scratch of its own: decoder.cc sizes that by this)
    if (global_tree) continue;
    for (uint32_t i = 0; i < nt; i++) { TransformDesc t; ReadTransform(r, &t); }
    FramePlan::LocalStream ls;
    ReadTree(r, &ls.tree, std::min<size_t>(1u << 20, 1024 + pixels));
    ReadEntropyCode(r, ls.tree.num_leaves, &ls.code);
    ls.unit = 1 + unit;
    ls.data_bitpos = r.pos();
    if (ls.tree.uses_wp) p->mod_local_wp = true;
    p->max_prop = std::max(p->max_prop, ls.tree.max_prop);
    p->local_streams.push_back(std::move(ls));
  }
}

void ParseHfGlobal(const Codestream& cs, const ImageHeader& ih, uint64_t bitpos, FramePlan* p) {
  (void)ih;
  Reader r(cs, bitpos);
  for (int k = 0; k < 17; k++) p->qspec[k] = QuantTableSpec();
  if (!r.b()) {
    for (int k = 0; k < 17; k++) {
      QuantTableSpec& q = p->qspec[k];
      q.mode = r.u(3);
      auto bands = [&](uint32_t* n, float (*v)[17]) {
        *n = r.u(4) + 1;
        for (int c = 0; c < 3; c++) for (uint32_t i = 0; i < *n; i++) v[c][i] = r.F16();
        for (int c = 0; c < 3; c++) v[c][0] *= 64.0f;
      };
      switch (q.mode) {
        case 0: break;
        case 1: if (k != 1) Fail("quant mode/table mismatch"); for (int c = 0; c < 3; c++) for (int i = 0; i < 3; i++) q.idw[c][i] = r.F16() * 64.0f; break;
        case 2: if (k != 2) Fail("quant mode/table mismatch"); for (int c = 0; c < 3; c++) for (int i = 0; i < 6; i++) q.dct2w[c][i] = r.F16() * 64.0f; break;
        case 3: if (k != 3) Fail("quant mode/table mismatch"); for (int c = 0; c < 3; c++) for (int i = 0; i < 2; i++) q.dct4mul[c][i] = r.F16(); bands(&q.num_bands, q.bands); break;
        case 4: if (k != 9) Fail("quant mode/table mismatch"); for (int c = 0; c < 3; c++) q.dct4x8mul[c] = r.F16(); bands(&q.num_bands, q.bands); break;
        case 5:
          if (k != 10) Fail("quant mode/table mismatch");
          for (int c = 0; c < 3; c++) for (int i = 0; i < 9; i++) { q.afvw[c][i] = r.F16(); if (i < 6) q.afvw[c][i] *= 64.0f; }
          bands(&q.num_bands, q.bands); bands(&q.num_bands4, q.bands4);
          break;
        case 6: bands(&q.num_bands, q.bands); break;
        case 7: {
          // RAW table: F16 denominator + a 3-channel Modular image (X, Y, B; libjxl coefficient layout) coded with the
          // global MA tree — a few hundred samples, decoded here with the same per-thread decoder the kernels use.
          q.raw_den = r.F16();
          const int rows = 8 * kKindRows[k], cols = 8 * kKindCols[k];
          if (!p->has_global_tree) Fail("RAW quant table without global tree");
          const bool use_global = r.b();
          WPHeader wp{16, 10, {7, 7, 7, 0, 0}, {13, 12, 12, 12}};
          if (!r.b()) { wp.p1 = r.u(5); wp.p2 = r.u(5); for (int i = 0; i < 5; i++) wp.p3[i] = r.u(5); for (int i = 0; i < 4; i++) wp.w[i] = r.u(4); }
          const uint32_t nt = r.U32({0, 0}, {0, 1}, {4, 2}, {8, 18});
          if (!use_global || nt != 0) Unsupported("RAW quant table with local tree / transforms");
          if (p->tree_code.lz77) Unsupported("RAW quant table in an LZ77-coded stream");
          const DevCode view = p->tree_code.View();
          ModularCtx mc;
          mc.tree = p->tree.nodes.data(); mc.code = &view; mc.wp = wp; mc.uses_wp = p->tree.uses_wp;
          vec<int32_t> wps(10 * (cols + 2), 0);
          mc.wp_scratch = wps.data();
          mc.stream_id = 1 + 3 * p->num_lf_groups + k;
          AnsReader ans; ans.Init(r.br, view);
          for (int c = 0; c < 3; c++) {
            q.raw[c].assign((size_t)rows * cols, 0);
            ChannelDesc ch; ch.data = q.raw[c].data(); ch.w = cols; ch.h = rows; ch.stride = cols;
            DecodeModularChannel(r.br, ans, mc, ch, c);
          }
          if (!ans.FinalOk(view)) Fail("RAW quant table ANS final state");
          if (r.pos() > r.limit_bits) throw ParseError("truncated", false);
          break;
        }
      }
    }
  }
  p->num_hf_presets = 1 + r.u(CeilLog2(p->num_groups));
  p->used_orders.assign(p->num_passes, 0);
  p->custom_order.assign((size_t)p->num_passes * 39, {});
  p->ac_code.resize(p->num_passes);
  for (uint32_t ps = 0; ps < p->num_passes; ps++) {
    uint32_t used = r.U32({0, 0x5F}, {0, 0x13}, {0, 0}, {13, 0});
    p->used_orders[ps] = used;
    if (used) {
      HostCode oc;
      ReadEntropyCode(r, 8, &oc);
      HostSymbolReader sr(r, oc);
      auto ctxof = [](uint32_t v) { uint32_t t = v == 0 ? 0 : 1 + FloorLog2(v); return std::min<uint32_t>(t, 7); };
      for (int b = 0; b < 13; b++) {
        if (!(used & (1u << b))) continue;
        vec<uint16_t> natural = NaturalCoeffOrder(kBucketStrategy[b]);
        const size_t size = natural.size(), skip = size / 64;
        for (int c = 0; c < 3; c++) {
          vec<uint32_t> lehmer(size, 0);
          uint32_t end = sr.Read(ctxof((uint32_t)size)) + (uint32_t)skip;
          if (end > size) Fail("permutation size");
          uint32_t last = 0;
          for (size_t i = skip; i < end; i++) { lehmer[i] = sr.Read(ctxof(last)); last = lehmer[i]; if (lehmer[i] >= size - i) Fail("lehmer code"); }
          vec<uint32_t> temp(size);
          for (size_t i = 0; i < size; i++) temp[i] = (uint32_t)i;
          vec<uint16_t>& o = p->custom_order[(size_t)ps * 39 + b * 3 + c];
          o.resize(size);
          for (size_t i = 0; i < size; i++) { o[i] = natural[temp[lehmer[i]]]; temp.erase(temp.begin() + lehmer[i]); }
        }
      }
      sr.CheckFinal();
    }
    ReadEntropyCode(r, 495 * p->bcm.num_ctxs * p->num_hf_presets, &p->ac_code[ps]);
  }
  p->end_bitpos = r.pos();
}

// ---- tables ---------------------------------------------------------------------------------------------------------------
const uint8_t kBucketStrategy[13] = {0, 1, 4, 5, 6, 8, 10, 18, 19, 21, 22, 24, 25};
const uint8_t kKindRows[17] = {1, 1, 1, 1, 2, 4, 1, 1, 2, 1, 1, 8, 4, 16, 8, 32, 16};
const uint8_t kKindCols[17] = {1, 1, 1, 1, 2, 4, 2, 4, 4, 1, 1, 8, 8, 16, 16, 32, 32};

vec<uint16_t> NaturalCoeffOrder(int strategy) {
  int cx = (int)CoveredX(strategy), cy = (int)CoveredY(strategy);
  if (cy > cx) std::swap(cx, cy);
  const int xs = cx * 8, ratio = cx / cy, mask = ratio - 1;
  int lr = 0; while ((1 << lr) < ratio) lr++;
  vec<uint16_t> out((size_t)cx * cy * 64);
  size_t cur = (size_t)cx * cy;
  auto emit = [&](int x, int y, bool first_half) {
    if (y & mask) return;
    y >>= lr;
    size_t k = (first_half && x < cx && y < cy) ? (size_t)y * cx + x : cur++;
    out[k] = (uint16_t)(y * xs + x);
  };
  for (int i = 0; i < xs; i++) for (int j = 0; j <= i; j++) { int x = j, y = i - j; if (i & 1) std::swap(x, y); emit(x, y, true); }
  for (int i = xs - 2; i >= 0; i--) for (int j = 0; j <= i; j++) { int x = xs - 1 - (i - j), y = xs - 1 - j; if (i & 1) std::swap(x, y); emit(x, y, false); }
  return out;
}

static QuantTableSpec LibrarySpec(int kind) {
  // quant_weights.cc library defaults [R]: only band-parameterised kinds are reproduced
  QuantTableSpec q;
  auto set = [&](int n, std::initializer_list<float> x, std::initializer_list<float> y, std::initializer_list<float> b) {
    q.mode = 6; q.num_bands = n;
    int i = 0; for (float v : x) q.bands[0][i++] = v;
    i = 0; for (float v : y) q.bands[1][i++] = v;
    i = 0; for (float v : b) q.bands[2][i++] = v;
  };
  switch (kind) {
    case 0: set(6, {3150.0f, 0.0f, -0.4f, -0.4f, -0.4f, -2.0f}, {560.0f, 0.0f, -0.3f, -0.3f, -0.3f, -0.3f}, {512.0f, -2.0f, -1.0f, 0.0f, -1.0f, -2.0f}); break;
    case 1: { q.mode = 1; float w[3][3] = {{280.0f, 3160.0f, 3160.0f}, {60.0f, 864.0f, 864.0f}, {18.0f, 200.0f, 200.0f}}; memcpy(q.idw, w, sizeof(w)); break; }
    case 2: { q.mode = 2; float w[3][6] = {{3840.0f, 2560.0f, 1280.0f, 640.0f, 480.0f, 300.0f}, {960.0f, 640.0f, 320.0f, 180.0f, 140.0f, 120.0f}, {640.0f, 320.0f, 128.0f, 64.0f, 32.0f, 16.0f}}; memcpy(q.dct2w, w, sizeof(w)); break; }
    case 3: set(4, {2200.0f, 0.0f, 0.0f, 0.0f}, {392.0f, 0.0f, 0.0f, 0.0f}, {112.0f, -0.25f, -0.25f, -0.5f}); q.mode = 3; for (int c = 0; c < 3; c++) q.dct4mul[c][0] = q.dct4mul[c][1] = 1.0f; break;
    case 4: set(7, {8996.8725711814115328f, -1.3000777393353804f, -0.49424529824571225f, -0.439093774457103443f, -0.6350101832695744f, -0.90177264050827612f, -1.6162099239887414f},
                {3191.48366296844234752f, -0.67424582104194355f, -0.80745813428471001f, -0.44925837484843441f, -0.35865440981033403f, -0.31322389111877305f, -0.37615025315725483f},
                {1157.50408145487200256f, -2.0531423165804414f, -1.4f, -0.50687130033378396f, -0.42708730624733904f, -1.4856834539296244f, -4.9209142884401604f}); break;
    case 5: set(8, {15718.40830982518931456f, -1.025f, -0.98f, -0.9012f, -0.4f, -0.48819395464f, -0.421064f, -0.27f},
                {7305.7636810695983104f, -0.8041958212306401f, -0.7633036457487539f, -0.55660379990111464f, -0.49785304658857626f, -0.43699592683512467f, -0.40180866526242109f, -0.27321683125358037f},
                {3803.53173721215041536f, -3.060733579805728f, -2.0413270132490346f, -2.0235650159727417f, -0.5495389509954993f, -0.4f, -0.4f, -0.3f}); break;
    case 6: set(7, {7240.7734393502f, -0.7f, -0.7f, -0.2f, -0.2f, -0.2f, -0.5f}, {1448.15468787004f, -0.5f, -0.5f, -0.5f, -0.2f, -0.2f, -0.2f}, {506.854140754517f, -1.4f, -0.2f, -0.5f, -0.5f, -1.5f, -3.6f}); break;
    case 7: set(8, {16283.2494710648897f, -1.7812845336559429f, -1.6309059012653515f, -1.0382179034313539f, -0.85f, -0.7f, -0.9f, -1.2360638576849587f},
                {5089.15750884921511936f, -0.320049391452786891f, -0.35362849922161446f, -0.30340000000000003f, -0.61f, -0.5f, -0.5f, -0.6f},
                {3397.77603275308720128f, -0.321327362693153371f, -0.34507619223117997f, -0.70340000000000003f, -0.9f, -1.0f, -1.0f, -1.1754605576265209f}); break;
    case 8: set(8, {13844.97076442300573f, -0.97113799999999995f, -0.658f, -0.42026f, -0.22712f, -0.2206f, -0.226f, -0.6f},
                {4798.964084220744293f, -0.61125308982767057f, -0.83770786552491361f, -0.79014862079498627f, -0.2692727459704829f, -0.38272769465388551f, -0.22924222653091453f, -0.20719098826199578f},
                {1807.236946760964614f, -1.2f, -1.2f, -0.7f, -0.7f, -0.7f, -0.4f, -0.5f}); break;
    case 9: set(4, {2198.050556016380522f, -0.96269623020744692f, -0.76194253026666783f, -0.6551140670773547f}, {764.3655248643528689f, -0.92630200888366945f, -0.9675229603596517f, -0.27845290869168118f},
                {527.107573587542228f, -1.4594385811273854f, -1.450082094097871593f, -1.5843722511996204f}); q.mode = 4; for (int c = 0; c < 3; c++) q.dct4x8mul[c] = 1.0f; break;
    case 10: {  // AFV: corner weights + the DCT4X8 and DCT4X4 band parameters
      const QuantTableSpec q48 = LibrarySpec(9), q44 = LibrarySpec(3);
      q.mode = 5;
      const float w[3][9] = {{3072.0f, 3072.0f, 256.0f, 256.0f, 256.0f, 414.0f, 0.0f, 0.0f, 0.0f}, {1024.0f, 1024.0f, 50.0f, 50.0f, 50.0f, 58.0f, 0.0f, 0.0f, 0.0f},
                             {384.0f, 384.0f, 12.0f, 12.0f, 12.0f, 22.0f, -0.25f, -0.25f, -0.25f}};
      memcpy(q.afvw, w, sizeof(w));
      q.num_bands = q48.num_bands; memcpy(q.bands, q48.bands, sizeof(q.bands));
      q.num_bands4 = q44.num_bands; memcpy(q.bands4, q44.bands, sizeof(q.bands4));
      break;
    }
    default: {
      static const float k64[3] = {26629.073922049845f, 9311.3238710010046f, 4992.2486445538634f}, k32[3] = {23629.073922049845f, 8611.3238710010046f, 4492.2486445538634f};
      float mul; const float* base;
      switch (kind) { case 11: mul = 0.9f; base = k64; break; case 12: mul = 0.65f; base = k32; break; case 13: mul = 1.8f; base = k64; break;
                      case 14: mul = 1.3f; base = k32; break; case 15: mul = 3.6f; base = k64; break; default: mul = 2.6f; base = k32; break; }
      set(8, {mul * base[0], -1.025f, -0.78f, -0.65012f, -0.19041574084286472f, -0.20819395464f, -0.421064f, -0.32733845535848671f},
          {mul * base[1], -0.3041958212306401f, -0.3633036457487539f, -0.35660379990111464f, -0.3443074455424403f, -0.33699592683512467f, -0.30180866526242109f, -0.27321683125358037f},
          {mul * base[2], -1.2f, -1.2f, -0.8f, -0.7f, -0.7f, -0.4f, -0.5f});
    }
  }
  return q;
}

// base/fast_math-inl.h FastLog2f / FastPow2f / FastPowf: the rational approximations libjxl interpolates quantisation bands with
// (quant_weights.cc InterpolateVec) — not std::pow, the tables differ by up to 3e-5 relative.
static float FastLog2f(float x) {
  int32_t x_bits; memcpy(&x_bits, &x, 4);
  const int32_t exp_bits = x_bits - 0x3f2aaaab;
  const int32_t exp_shifted = exp_bits >> 23;
  const int32_t m_bits = x_bits - (int32_t)((uint32_t)exp_shifted << 23);
  float mantissa; memcpy(&mantissa, &m_bits, 4);
  const float t = mantissa - 1.0f;
  float yp = std::fmaf(7.4245873327820566E-01f, t, 1.4287160470083755E+00f); yp = std::fmaf(yp, t, -1.8503833400518310E-06f);
  float yq = std::fmaf(1.7409343003366853E-01f, t, 1.0096718572241148E+00f); yq = std::fmaf(yq, t, 9.9032814277590719E-01f);
  return yp / yq + (float)exp_shifted;
}
static float FastPow2f(float x) {
  const float floorx = std::floor(x);
  const int32_t e_bits = (int32_t)((uint32_t)((int32_t)floorx + 127) << 23);
  float exp; memcpy(&exp, &e_bits, 4);
  const float frac = x - floorx;
  float num = frac + 1.01749063e+01f;
  num = std::fmaf(num, frac, 4.88687798e+01f);
  num = std::fmaf(num, frac, 9.85506591e+01f);
  num = num * exp;
  float den = std::fmaf(frac, 2.10242958e-01f, -2.22328856e-02f);
  den = std::fmaf(den, frac, -1.94414990e+01f);
  den = std::fmaf(den, frac, 9.85506633e+01f);
  return num / den;
}
float FastPowf(float base, float exponent) { return FastPow2f(FastLog2f(base) * exponent); }

static void BandWeights(uint32_t nb, const float (*bands_in)[17], int c, int ROWS, int COLS, float* out) {
  float bands[17];
  bands[0] = bands_in[c][0];
  if (!(bands[0] >= 1e-8f)) Fail("quant band");
  for (uint32_t i = 1; i < nb; i++) {
    float v = bands_in[c][i];
    bands[i] = bands[i - 1] * (v > 0 ? 1.0f + v : 1.0f / (1.0f - v));
    if (!(bands[i] >= 1e-8f)) Fail("quant band");
  }
  const float kSqrt2 = 1.41421356237309504880f;
  float scale = (nb - 1) / (kSqrt2 + 1e-6f), rcpcol = scale / (COLS - 1), rcprow = scale / (ROWS - 1);
  for (int y = 0; y < ROWS; y++) {
    float dy = y * rcprow, dy2 = dy * dy;
    for (int x = 0; x < COLS; x++) {
      float dx = x * rcpcol;
      float dist = std::sqrt(std::fmaf(dx, dx, dy2));
      float w;
      if (nb == 1) w = bands[0];
      else {
        int idx = (int)dist;
        if (idx + 1 >= (int)nb) idx = (int)nb - 2;
        float frac = dist - idx;
        w = bands[idx] * FastPowf(bands[idx + 1] / bands[idx], frac);
      }
      out[y * COLS + x] = w;
    }
  }
}

void ComputeQuantTable(const QuantTableSpec& spec0, int kind, int c, vec<float>* out) {
  QuantTableSpec lib;
  const QuantTableSpec* q = &spec0;
  if (spec0.mode == 0) { lib = LibrarySpec(kind); q = &lib; }
  const int ROWS = 8 * kKindRows[kind], COLS = 8 * kKindCols[kind];
  const size_t n = (size_t)ROWS * COLS;
  vec<float> w(n, 1.0f);
  switch (q->mode) {
    case 6: BandWeights(q->num_bands, q->bands, c, ROWS, COLS, w.data()); break;
    case 1: for (auto& v : w) v = q->idw[c][0]; w[1] = w[8] = q->idw[c][1]; w[9] = q->idw[c][2]; break;
    case 2: {
      const float* d = q->dct2w[c];
      w[0] = 1e6f; w[1] = w[8] = d[0]; w[9] = d[1];
      for (int y = 0; y < 2; y++) for (int x = 0; x < 2; x++) { w[y * 8 + x + 2] = d[2]; w[(y + 2) * 8 + x] = d[2]; w[(y + 2) * 8 + x + 2] = d[3]; }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { w[y * 8 + x + 4] = d[4]; w[(y + 4) * 8 + x] = d[4]; w[(y + 4) * 8 + x + 4] = d[5]; }
      break;
    }
    case 3: {
      float w4[16]; BandWeights(q->num_bands, q->bands, c, 4, 4, w4);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[y * 8 + x] = w4[(y / 2) * 4 + x / 2];
      w[1] /= q->dct4mul[c][0]; w[8] /= q->dct4mul[c][0]; w[9] /= q->dct4mul[c][1];
      break;
    }
    case 4: {
      float w48[32]; BandWeights(q->num_bands, q->bands, c, 4, 8, w48);
      for (int y = 0; y < 8; y++) for (int x = 0; x < 8; x++) w[y * 8 + x] = w48[(y / 2) * 8 + x];
      w[8] /= q->dct4x8mul[c];
      break;
    }
    case 5: {  // quant_weights.cc kQuantModeAFV
      static const float kFreqs[16] = {0, 0, 0.8517778890324296f, 5.37778436506804f, 0, 0, 4.734747904497923f, 5.449245381693219f,
                                       1.6598270267479331f, 4.0f, 7.275749096817861f, 10.423227632456525f, 2.662932286148962f, 7.630657783650829f,
                                       8.962388608184032f, 12.97166202570235f};
      float w48[32], w44[16];
      BandWeights(q->num_bands, q->bands, c, 4, 8, w48);
      BandWeights(q->num_bands4, q->bands4, c, 4, 4, w44);
      const float lo = 0.8517778890324296f, hi = 12.97166202570235f - lo + 1e-6f;
      float bands[4];
      bands[0] = q->afvw[c][5];
      if (!(bands[0] >= 1e-8f)) Fail("AFV quant band");
      for (int i = 1; i < 4; i++) { const float v = q->afvw[c][i + 5]; bands[i] = bands[i - 1] * (v > 0 ? 1.0f + v : 1.0f / (1.0f - v)); if (!(bands[i] >= 1e-8f)) Fail("AFV quant band"); }
      w[0] = 1.0f;
      w[1 * 8 + 0] = q->afvw[c][0]; w[0 * 8 + 1] = q->afvw[c][1];
      w[2 * 8 + 0] = q->afvw[c][2]; w[0 * 8 + 2] = q->afvw[c][3]; w[2 * 8 + 2] = q->afvw[c][4];
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) {
        if (x < 2 && y < 2) continue;
        const float scaled_pos = (kFreqs[y * 4 + x] - lo) * 3 / hi;
        const int idx = (int)scaled_pos;
        w[(2 * y) * 8 + 2 * x] = bands[idx] * FastPowf(bands[idx + 1] / bands[idx], scaled_pos - idx);
      }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 8; x++) { if (x == 0 && y == 0) continue; w[(2 * y + 1) * 8 + x] = w48[y * 8 + x]; }
      for (int y = 0; y < 4; y++) for (int x = 0; x < 4; x++) { if (x == 0 && y == 0) continue; w[(2 * y) * 8 + 2 * x + 1] = w44[y * 4 + x]; }
      break;
    }
    case 7:
      if (q->raw[c].size() != n) Fail("RAW quant table size");
      out->resize(n);
      for (size_t i = 0; i < n; i++) { if (q->raw[c][i] <= 0) Fail("RAW quant value"); (*out)[i] = 1.0f / (1.0f / (q->raw_den * (float)q->raw[c][i])); }
      return;
    default: Fail("quant mode");
  }
  out->resize(n);
  for (size_t i = 0; i < n; i++) { if (!(w[i] > 0) || !std::isfinite(w[i])) Fail("quant weight"); (*out)[i] = 1.0f / w[i]; }
}

}  // namespace jxlhip
