// jxl-hip: jbrd parsing and JPEG serialisation (see jpeg_recon.h).  Field layout: libjxl lib/jxl/jpeg/jpeg_data.cc JPEGData::VisitFields;
// payload layout: dec_jpeg_data.cc DecodeJPEGData; writer: dec_jpeg_data_writer.cc.  Checked on the reference's fixture: the jbrd box of
// samples/sample_jpg.jxl parses to the marker order, Huffman tables and APP0 payload of samples/sample.jpg, and the written file is
// byte-identical to it (tests/test_abi_host.py, tests/test_gpu_parity.py).
#include "jpeg_recon.h"
#include <dlfcn.h>
#include <algorithm>
#include <cstring>
#include <mutex>

namespace jxlhip {

namespace {

struct Bits {   // LSB-first bit reader (fields.h / dec_bit_reader.h)
  const uint8_t* d; size_t n; size_t p = 0; bool bad = false;
  Bits(const uint8_t* data, size_t size) : d(data), n(size) {}
  uint32_t u(int k) {
    uint32_t v = 0;
    for (int i = 0; i < k; i++) {
      if ((p >> 3) >= n) { bad = true; return 0; }
      v |= (uint32_t)((d[p >> 3] >> (p & 7)) & 1) << i;
      p++;
    }
    return v;
  }
  bool b() { return u(1) != 0; }
  struct D { int bits; uint32_t off; };
  uint32_t U32(D d0, D d1, D d2, D d3) { const uint32_t s = u(2); const D k = s == 0 ? d0 : s == 1 ? d1 : s == 2 ? d2 : d3; return k.off + u(k.bits); }
};

typedef int (*BrotliDecompressFn)(size_t, const uint8_t*, size_t*, uint8_t*);
BrotliDecompressFn LoadBrotli() {
  static std::once_flag once;
  static BrotliDecompressFn fn = nullptr;
  std::call_once(once, [] {
    for (const char* name : {"libbrotlidec.so.1", "libbrotlidec.so"}) {
      if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) { fn = (BrotliDecompressFn)dlsym(h, "BrotliDecoderDecompress"); if (fn) break; }
    }
  });
  return fn;
}

}  // namespace
bool BrotliDecompressAll(const uint8_t* data, size_t size, size_t limit, vec<uint8_t>* out) {
  BrotliDecompressFn brotli = LoadBrotli();
  if (!brotli) return false;
  for (size_t cap = std::max<size_t>(4096, size * 4); ; cap *= 4) {
    if (cap > limit) cap = limit;
    out->resize(cap);
    size_t got = cap;
    if (brotli(size, data, &got, out->data()) == 1) { out->resize(got); return true; }     // (the one-shot API reports "output full" and "damaged" alike)
    if (cap >= limit) { out->clear(); return false; }
  }
}
namespace {
const uint8_t kIccTag[12] = {'I', 'C', 'C', '_', 'P', 'R', 'O', 'F', 'I', 'L', 'E', 0};
const uint8_t kExifTag[6] = {'E', 'x', 'i', 'f', 0, 0};
const uint8_t kXmpTag[29] = {'h', 't', 't', 'p', ':', '/', '/', 'n', 's', '.', 'a', 'd', 'o', 'b', 'e', '.', 'c', 'o', 'm', '/', 'x', 'a', 'p', '/', '1', '.', '0', '/', 0};

const uint8_t kNaturalOrder[64] = {0,  1,  8,  16, 9,  2,  3,  10, 17, 24, 32, 25, 18, 11, 4,  5,  12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6,  7,  14, 21, 28,
                                   35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

}  // namespace

bool ParseJbrd(const uint8_t* data, size_t size, JpegData* jd, std::string* err) {
  auto fail = [&](const char* m) { if (err) *err = std::string("jbrd: ") + m; return false; };
  *jd = JpegData();
  Bits r(data, size);
  const bool is_gray = r.b();
  jd->components.resize(is_gray ? 1 : 3);
  size_t num_app = 0, num_com = 0, num_scans = 0, num_inter = 0;
  bool has_dri = false;
  for (;;) {
    const uint8_t m = (uint8_t)(r.u(6) + 0xC0);
    if (r.bad) return fail("truncated marker list");
    if ((m & 0xF0) == 0xE0) num_app++;
    if (m == 0xFE) num_com++;
    if (m == 0xDA) num_scans++;
    if (m == 0xFF) num_inter++;
    if (m == 0xDD) has_dri = true;
    jd->marker_order.push_back(m);
    if (jd->marker_order.size() > 16384) return fail("too many markers");
    if (m == 0xD9) break;
  }
  jd->app_data.resize(num_app); jd->app_marker_type.resize(num_app); jd->com_data.resize(num_com); jd->scan_info.resize(num_scans);
  for (size_t i = 0; i < num_app; i++) {
    jd->app_marker_type[i] = r.U32({0, 0}, {0, 1}, {1, 2}, {2, 4});
    if (jd->app_marker_type[i] > 3) return fail("unknown app marker type");
    jd->app_data[i].resize((size_t)r.u(16) + 1);
    if (jd->app_data[i].size() < 3) return fail("invalid marker size");
  }
  for (auto& c : jd->com_data) { c.resize((size_t)r.u(16) + 1); if (c.size() < 3) return fail("invalid marker size"); }
  const uint32_t nq = r.U32({0, 1}, {0, 2}, {0, 3}, {0, 4});
  if (nq == 4) return fail("invalid number of quant tables");
  jd->quant.resize(nq);
  for (auto& q : jd->quant) { q.precision = r.u(1); q.index = r.u(2); q.is_last = r.b(); }
  const uint32_t ctype = r.u(2);   // 0 gray, 1 YCbCr, 2 RGB, 3 custom
  uint32_t ncomp = ctype == 0 ? 1 : 3;
  if (ctype == 3) { ncomp = r.U32({0, 1}, {0, 2}, {0, 3}, {0, 4}); if (ncomp != 1 && ncomp != 3) return fail("invalid number of components"); }
  jd->components.resize(ncomp);
  if (ctype == 3) for (auto& c : jd->components) c.id = r.u(8);
  else if (ctype == 0) jd->components[0].id = 1;
  else if (ctype == 2) { jd->components[0].id = 'R'; jd->components[1].id = 'G'; jd->components[2].id = 'B'; }
  else { jd->components[0].id = 1; jd->components[1].id = 2; jd->components[2].id = 3; }
  for (auto& c : jd->components) { c.quant_idx = r.u(2); if (c.quant_idx >= jd->quant.size()) return fail("invalid quant table index"); }
  const uint32_t nh = r.U32({0, 4}, {3, 2}, {4, 10}, {6, 26});
  jd->huffman_code.resize(nh);
  for (auto& h : jd->huffman_code) {
    const bool is_ac = r.b();
    const uint32_t id = r.u(2);
    h.slot_id = ((uint32_t)is_ac << 4) | id;
    h.is_last = r.b();
    size_t nsym = 0;
    for (int i = 0; i <= 16; i++) { h.counts[i] = r.U32({0, 0}, {0, 1}, {3, 2}, {8, 0}); nsym += h.counts[i]; }
    if (nsym < 1 || nsym > 257) return fail("bad Huffman table size");
    h.values.resize(nsym);
    for (auto& v : h.values) v = r.U32({2, 0}, {2, 4}, {4, 8}, {8, 1});
    if (h.values.back() != 256) return fail("missing EOI symbol");
    if (r.bad) return fail("truncated");
  }
  for (auto& s : jd->scan_info) {
    s.num_components = r.U32({0, 1}, {0, 2}, {0, 3}, {0, 4});
    if (s.num_components >= 4) return fail("invalid number of components in SOS");
    s.Ss = r.u(6); s.Se = r.u(6); s.Al = r.u(4); s.Ah = r.u(4);
    for (uint32_t i = 0; i < s.num_components; i++) {
      s.components[i].comp_idx = r.u(2);
      if (s.components[i].comp_idx >= jd->components.size()) return fail("invalid component index in SOS");
      s.components[i].ac_tbl_idx = r.u(2); s.components[i].dc_tbl_idx = r.u(2);
    }
    s.last_needed_pass = r.U32({0, 0}, {0, 1}, {0, 2}, {3, 3});
  }
  if (has_dri) jd->restart_interval = r.u(16);
  for (auto& s : jd->scan_info) {
    const uint32_t nrp = r.U32({0, 0}, {2, 1}, {4, 4}, {16, 20});
    s.reset_points.resize(nrp);
    int64_t last = -1;
    for (auto& bidx : s.reset_points) {
      bidx = r.U32({0, 0}, {3, 1}, {5, 9}, {28, 41}) + (uint32_t)(last + 1);
      if (bidx >= (3u << 26)) return fail("invalid block id");
      last = bidx;
    }
    const uint32_t nez = r.U32({0, 0}, {2, 1}, {4, 4}, {16, 20});
    s.extra_zero_runs.resize(nez);
    last = -1;
    for (auto& e : s.extra_zero_runs) {
      e.second = r.U32({0, 1}, {2, 2}, {4, 5}, {8, 20});
      e.first = r.U32({0, 0}, {3, 1}, {5, 9}, {28, 41}) + (uint32_t)(last + 1);
      if (e.first > (3u << 26)) return fail("invalid block id");
      last = e.first;
    }
    if (r.bad) return fail("truncated");
  }
  vec<uint32_t> inter_sizes(num_inter);
  for (auto& v : inter_sizes) v = r.u(16);
  const uint32_t tail_len = r.U32({0, 0}, {8, 1}, {16, 257}, {22, 65793});
  jd->has_zero_padding_bit = r.b();
  if (jd->has_zero_padding_bit) {
    const uint32_t nbit = r.u(24);
    if ((uint64_t)nbit > (uint64_t)size * 8) return fail("padding bits");
    jd->padding_bits.resize(nbit);
    for (auto& b : jd->padding_bits) b = (uint8_t)r.u(1);
  }
  if (r.bad) return fail("truncated");
  {
    // the announced sizes are checked against what the Brotli stream behind them can plausibly hold before anything of that size is
    // allocated (a 45 KB box may announce 1 GiB of inter-marker data): three orders of magnitude of compression is beyond any JPEG's markers
    uint64_t announced = tail_len;
    for (uint32_t v : inter_sizes) announced += v;
    const size_t off0 = (r.p + 7) / 8;
    const uint64_t brotli_bytes = off0 < size ? size - off0 : 0;
    // (Brotli reaches ratios far beyond 1000:1 on zero-filled tails, so no ratio is assumed: an absolute bound on what is allocated, and the
    // decompressed size has to match exactly below)
    if (announced > ((uint64_t)1 << 30) || (announced > 0 && brotli_bytes == 0)) return fail("announced marker data exceeds what the box can hold");
  }
  jd->tail_data.resize(tail_len);
  for (uint32_t v : inter_sizes) jd->inter_marker_data.emplace_back(v);
  // ---- Brotli stream: unknown-type APPn markers, COM markers, inter-marker data, tail data, back to back
  size_t total = 0;
  for (size_t i = 0; i < num_app; i++) if (jd->app_marker_type[i] == 0) total += jd->app_data[i].size();
  for (auto& c : jd->com_data) total += c.size();
  for (auto& d : jd->inter_marker_data) total += d.size();
  total += jd->tail_data.size();
  const size_t off = (r.p + 7) / 8;
  vec<uint8_t> plain(total + 1);
  if (total > 0) {
    BrotliDecompressFn brotli = LoadBrotli();
    if (!brotli) return fail("libbrotlidec.so.1 not available");
    size_t got = plain.size();
    if (off >= size || brotli(size - off, data + off, &got, plain.data()) != 1 || got != total) return fail("Brotli stream does not match the announced sizes");
  }
  size_t pos = 0;
  auto take = [&](vec<uint8_t>& v) { if (!v.empty()) memcpy(v.data(), plain.data() + pos, v.size()); pos += v.size(); };
  uint32_t num_icc = 0;
  for (size_t i = 0; i < num_app; i++) {
    auto& a = jd->app_data[i];
    const uint32_t type = jd->app_marker_type[i];
    if (type == 0) { take(a); if ((size_t)a[1] * 256u + a[2] + 1u != a.size()) return fail("APP marker length mismatch"); continue; }
    // dec_jpeg_data.cc: marker byte, length and tag of the markers whose payload lives elsewhere in the file
    const size_t tag_len = type == 1 ? sizeof(kIccTag) : type == 2 ? sizeof(kExifTag) : sizeof(kXmpTag);
    if (a.size() < 3 + tag_len + (type == 1 ? 2 : 0)) return fail("metadata marker too short");
    a[0] = type == 1 ? 0xE2 : 0xE1;
    a[1] = (uint8_t)((a.size() - 1) >> 8); a[2] = (uint8_t)(a.size() - 1);
    memcpy(&a[3], type == 1 ? kIccTag : type == 2 ? kExifTag : kXmpTag, tag_len);
    if (type == 1) a[15] = (uint8_t)++num_icc;
  }
  for (size_t i = 0; i < num_app; i++) if (jd->app_marker_type[i] == 1) jd->app_data[i][16] = (uint8_t)num_icc;
  for (auto& c : jd->com_data) { take(c); if ((size_t)c[1] * 256u + c[2] + 1u != c.size()) return fail("COM marker length mismatch"); }
  for (auto& d : jd->inter_marker_data) take(d);
  take(jd->tail_data);
  return true;
}

bool FillJpegMetadata(JpegData* jd, const JpegMetadataSources& src, std::string* err) {
  auto fail = [&](const char* m) { if (err) *err = std::string("JPEG metadata: ") + m; return false; };
  // a box payload of known size, through Brotli when it came in a `brob` box
  auto payload = [&](const uint8_t* data, size_t size, bool brob, size_t want, vec<uint8_t>* out) {
    if (!brob) { if (size != want) return false; out->assign(data, data + size); return true; }
    BrotliDecompressFn brotli = LoadBrotli();
    if (!brotli) return false;
    out->resize(want + 1);
    size_t got = out->size();
    if (brotli(size, data, &got, out->data()) != 1 || got != want) return false;
    out->resize(want);
    return true;
  };
  size_t icc_pos = 0;
  bool exif_done = false, xmp_done = false;
  for (size_t i = 0; i < jd->app_data.size(); i++) {
    auto& a = jd->app_data[i];
    const uint32_t type = jd->app_marker_type[i];
    if (type == 1) {
      const size_t len = a.size() - 17;
      if (icc_pos + len > src.icc_size) return fail("ICC profile shorter than the APP2 markers");
      memcpy(a.data() + 17, src.icc + icc_pos, len);
      icc_pos += len;
    } else if (type == 2 && !exif_done) {
      if (!src.exif) return fail("Exif marker without an Exif box");
      const size_t want = a.size() - 3 - sizeof(kExifTag) + 4;     // the box starts with the 4-byte TIFF header offset
      vec<uint8_t> box;
      if (!payload(src.exif, src.exif_size, src.exif_brob, want, &box)) return fail("Exif size mismatch");
      memcpy(a.data() + 3 + sizeof(kExifTag), box.data() + 4, want - 4);
      exif_done = true;
    } else if (type == 3 && !xmp_done) {
      if (!src.xml) return fail("XMP marker without an xml box");
      const size_t want = a.size() - 3 - sizeof(kXmpTag);
      vec<uint8_t> box;
      if (!payload(src.xml, src.xml_size, src.xml_brob, want, &box)) return fail("XMP size mismatch");
      if (want) memcpy(a.data() + 3 + sizeof(kXmpTag), box.data(), want);
      xmp_done = true;
    } else if (type != 0) return fail("more than one Exif / XMP marker");
  }
  if (icc_pos != src.icc_size && icc_pos != 0) return fail("ICC profile longer than the APP2 markers");
  return true;
}

namespace {

struct HuffTable { uint8_t depth[256]; uint16_t code[256]; bool init = false; };

// dec_jpeg_data_writer.cc BuildHuffmanCodeTable: canonical JPEG code from counts / values (the 256 sentinel keeps the all-ones code free)
bool BuildHuffTable(const JpegHuffmanCode& h, HuffTable* t) {
  memset(t->depth, 127, sizeof(t->depth));
  memset(t->code, 0, sizeof(t->code));
  uint32_t code = 0, k = 0;
  for (int len = 1; len <= 16; len++) {
    for (uint32_t i = 0; i < h.counts[len]; i++) {
      if (k >= h.values.size()) return false;
      const uint32_t v = h.values[k++];
      if (v < 256) { t->depth[v] = (uint8_t)len; t->code[v] = (uint16_t)code; }
      code++;
    }
    code <<= 1;
  }
  t->init = true;
  return k == h.values.size();
}

struct BitWriter {
  vec<uint8_t>* out;
  uint64_t acc = 0; int nbits = 0;   // bits pending (MSB-first)
  bool ok = true;
  void Put(uint32_t v, int n) {
    acc = (acc << n) | (v & ((1u << n) - 1u)); nbits += n;
    while (nbits >= 8) {
      const uint8_t b = (uint8_t)(acc >> (nbits - 8));
      out->push_back(b);
      if (b == 0xFF) out->push_back(0);   // byte stuffing
      nbits -= 8;
    }
  }
  void Symbol(int sym, const HuffTable& t) { if (t.depth[sym] == 127) { ok = false; return; } Put(t.code[sym], t.depth[sym]); }
  // JumpToByteBoundary: pad with ones, or with the recorded padding bits
  bool Pad(const JpegData& jd, size_t* pad_pos) {
    const int n = (8 - (nbits & 7)) & 7;
    if (n == 0) return true;
    uint32_t pattern;
    if (!jd.has_zero_padding_bit) pattern = (1u << n) - 1;
    else {
      pattern = 0;
      for (int i = 0; i < n; i++) { if (*pad_pos >= jd.padding_bits.size()) return false; pattern = (pattern << 1) | (jd.padding_bits[(*pad_pos)++] ? 1u : 0u); }
    }
    Put(pattern, n);
    return true;
  }
};

// dec_jpeg_data_writer.cc DCTCodingState: the end-of-band run and the correction bits waiting for it (progressive scans)
struct EobState {
  uint32_t eob_run = 0;
  const HuffTable* ac = nullptr;
  vec<uint8_t> bits;
  void Flush(BitWriter& w) {
    if (eob_run > 0) {
      int nbits = 0;
      while ((eob_run >> (nbits + 1)) != 0) nbits++;
      w.Symbol(nbits << 4, *ac);
      if (nbits > 0) w.Put(eob_run & ((1u << nbits) - 1), nbits);
      eob_run = 0;
    }
    for (uint8_t b : bits) w.Put(b, 1);
    bits.clear();
  }
  void BufferEndOfBand(BitWriter& w, const HuffTable* table, const vec<uint8_t>* new_bits) {
    if (eob_run == 0) ac = table;
    eob_run++;
    if (new_bits) bits.insert(bits.end(), new_bits->begin(), new_bits->end());
    if (eob_run == 0x7FFF || bits.size() > (1u << 16) - 64 + 1) Flush(w);
  }
};

// EncodeDCTBlockProgressive: first pass over a spectral band at precision Al (DC difference when the band starts at 0)
bool EncodeBlockProgressive(const int16_t* c, const HuffTable& dct, const HuffTable& act, int Ss, int Se, int Al, int num_zero_runs, EobState& st, int* last_dc, BitWriter& w) {
  const bool eob_run_allowed = Ss > 0;
  if (Ss == 0) {
    int temp2 = c[0] >> Al, temp = temp2 - *last_dc;
    *last_dc = temp2;
    temp2 = temp;
    if (temp < 0) { temp = -temp; temp2--; }
    int nbits = 0;
    while ((temp >> nbits) != 0) nbits++;
    if (nbits >= 13) return false;
    w.Symbol(nbits, dct);
    if (nbits > 0) w.Put((uint32_t)temp2 & ((1u << nbits) - 1), nbits);
    Ss++;
  }
  if (Ss > Se) return true;
  int r = 0;
  for (int k = Ss; k <= Se; k++) {
    int temp = c[kNaturalOrder[k]], temp2;
    if (temp == 0) { r++; continue; }
    if (temp < 0) { temp = -temp; temp >>= Al; temp2 = ~temp; } else { temp >>= Al; temp2 = temp; }
    if (temp == 0) { r++; continue; }
    st.Flush(w);
    while (r > 15) { w.Symbol(0xF0, act); r -= 16; }
    int nbits = 0;
    while ((temp >> nbits) != 0) nbits++;
    if (nbits >= 16) return false;
    w.Symbol((r << 4) + nbits, act);
    w.Put((uint32_t)temp2 & ((1u << nbits) - 1), nbits);
    r = 0;
  }
  if (num_zero_runs > 0) {
    st.Flush(w);
    for (int i = 0; i < num_zero_runs; i++) { w.Symbol(0xF0, act); r -= 16; }
  }
  if (r > 0) {
    st.BufferEndOfBand(w, &act, nullptr);
    if (!eob_run_allowed) st.Flush(w);
  }
  return true;
}

// EncodeRefinementBits: one more bit of precision for a band (successive approximation, Ah > 0)
bool EncodeBlockRefinement(const int16_t* c, const HuffTable& act, int Ss, int Se, int Al, EobState& st, BitWriter& w) {
  const bool eob_run_allowed = Ss > 0;
  if (Ss == 0) { w.Put((uint32_t)(c[0] >> Al) & 1u, 1); Ss++; }
  if (Ss > Se) return true;
  int abs_values[64];
  int eob = 0;
  for (int k = Ss; k <= Se; k++) {
    const int v = c[kNaturalOrder[k]];
    abs_values[k] = (v < 0 ? -v : v) >> Al;
    if (abs_values[k] == 1) eob = k;
  }
  int r = 0;
  vec<uint8_t> refinement;
  refinement.reserve(64);
  for (int k = Ss; k <= Se; k++) {
    if (abs_values[k] == 0) { r++; continue; }
    while (r > 15 && k <= eob) {
      st.Flush(w);
      w.Symbol(0xF0, act);
      r -= 16;
      for (uint8_t b : refinement) w.Put(b, 1);
      refinement.clear();
    }
    if (abs_values[k] > 1) { refinement.push_back((uint8_t)(abs_values[k] & 1)); continue; }
    st.Flush(w);
    w.Symbol((r << 4) + 1, act);
    w.Put(c[kNaturalOrder[k]] < 0 ? 0u : 1u, 1);
    for (uint8_t b : refinement) w.Put(b, 1);
    refinement.clear();
    r = 0;
  }
  if (r > 0 || !refinement.empty()) {
    st.BufferEndOfBand(w, &act, &refinement);
    if (!eob_run_allowed) st.Flush(w);
  }
  return true;
}

}  // namespace

bool WriteJpeg(const JpegData& jd, uint32_t width, uint32_t height, const int16_t* const* coeffs, vec<uint8_t>* out, std::string* err) {
  auto fail = [&](const char* m) { if (err) *err = std::string("JPEG writer: ") + m; return false; };
  out->clear();
  out->push_back(0xFF); out->push_back(0xD8);
  size_t app_i = 0, com_i = 0, inter_i = 0, dqt_i = 0, dht_i = 0, scan_i = 0, pad_pos = 0;
  HuffTable dc_tab[4], ac_tab[4];
  // component planes: MCU grid x sampling factors (jpeg_data.h JPEGComponent::width_in_blocks / height_in_blocks)
  uint32_t max_h = 1, max_v = 1;
  for (auto& c : jd.components) { if (c.h_samp < 1 || c.h_samp > 4 || c.v_samp < 1 || c.v_samp > 4) return fail("bad sampling factor"); max_h = std::max(max_h, c.h_samp); max_v = std::max(max_v, c.v_samp); }
  const uint32_t mcu_cols = (width + 8 * max_h - 1) / (8 * max_h), mcu_rows = (height + 8 * max_v - 1) / (8 * max_v);
  bool seen_dri = false, is_progressive = false;
  for (uint8_t m : jd.marker_order) {
    if (m == 0xC0 || m == 0xC1 || m == 0xC2 || m == 0xC9 || m == 0xCA) {
      if (m == 0xC9 || m == 0xCA) return fail("unsupported: arithmetic-coded JPEG");
      is_progressive = m == 0xC2;
      const size_t n = jd.components.size(), len = 8 + 3 * n;
      const uint8_t hdr[9] = {0xFF, m, (uint8_t)(len >> 8), (uint8_t)len, 8, (uint8_t)(height >> 8), (uint8_t)height, (uint8_t)(width >> 8), (uint8_t)width};
      out->insert(out->end(), hdr, hdr + 9);
      out->push_back((uint8_t)n);
      for (auto& c : jd.components) { out->push_back((uint8_t)c.id); out->push_back((uint8_t)((c.h_samp << 4) | c.v_samp)); out->push_back((uint8_t)jd.quant[c.quant_idx].index); }
    } else if (m == 0xC4) {
      size_t len = 2, last = dht_i;
      for (size_t i = dht_i; i < jd.huffman_code.size(); i++) { len += 16; for (uint32_t c : jd.huffman_code[i].counts) len += c; last = i; if (jd.huffman_code[i].is_last) break; }
      if (dht_i >= jd.huffman_code.size()) return fail("DHT marker without tables");
      out->push_back(0xFF); out->push_back(0xC4); out->push_back((uint8_t)(len >> 8)); out->push_back((uint8_t)len);
      for (; dht_i <= last; dht_i++) {
        const JpegHuffmanCode& h = jd.huffman_code[dht_i];
        HuffTable* t = (h.slot_id & 0x10) ? &ac_tab[h.slot_id & 3] : &dc_tab[h.slot_id & 3];
        if ((h.slot_id & 0xF) > 3 || !BuildHuffTable(h, t)) return fail("bad Huffman table");
        size_t total = 0, max_len = 0;
        for (int i = 0; i <= 16; i++) { if (h.counts[i]) max_len = i; total += h.counts[i]; }
        total--;
        out->push_back((uint8_t)h.slot_id);
        for (size_t i = 1; i <= 16; i++) out->push_back((uint8_t)(i == max_len ? h.counts[i] - 1 : h.counts[i]));
        for (size_t i = 0; i < total; i++) out->push_back((uint8_t)h.values[i]);
      }
    } else if (m == 0xDB) {
      size_t len = 2, last = dqt_i;
      if (dqt_i >= jd.quant.size()) return fail("DQT marker without tables");
      for (size_t i = dqt_i; i < jd.quant.size(); i++) { len += 1 + (jd.quant[i].precision ? 128 : 64); last = i; if (jd.quant[i].is_last) break; }
      out->push_back(0xFF); out->push_back(0xDB); out->push_back((uint8_t)(len >> 8)); out->push_back((uint8_t)len);
      for (; dqt_i <= last; dqt_i++) {
        const JpegQuantTable& q = jd.quant[dqt_i];
        out->push_back((uint8_t)((q.precision << 4) | q.index));
        for (int i = 0; i < 64; i++) { const int v = q.values[kNaturalOrder[i]]; if (q.precision) out->push_back((uint8_t)(v >> 8)); out->push_back((uint8_t)v); }
      }
    } else if (m == 0xDD) {
      seen_dri = true;
      const uint8_t d[6] = {0xFF, 0xDD, 0, 4, (uint8_t)(jd.restart_interval >> 8), (uint8_t)jd.restart_interval};
      out->insert(out->end(), d, d + 6);
    } else if ((m & 0xF0) == 0xE0) {
      if (app_i >= jd.app_data.size()) return fail("APP marker without data");
      out->push_back(0xFF);
      out->insert(out->end(), jd.app_data[app_i].begin(), jd.app_data[app_i].end());
      app_i++;
    } else if (m == 0xFE) {
      if (com_i >= jd.com_data.size()) return fail("COM marker without data");
      out->push_back(0xFF);
      out->insert(out->end(), jd.com_data[com_i].begin(), jd.com_data[com_i].end());
      com_i++;
    } else if (m == 0xFF) {
      if (inter_i >= jd.inter_marker_data.size()) return fail("inter-marker data missing");
      out->insert(out->end(), jd.inter_marker_data[inter_i].begin(), jd.inter_marker_data[inter_i].end());
      inter_i++;
    } else if (m == 0xDA) {
      if (scan_i >= jd.scan_info.size()) return fail("SOS marker without scan info");
      const JpegScanInfo& s = jd.scan_info[scan_i++];
      if (s.Ss > s.Se || s.Se > 63 || s.Al > 13 || s.Ah > 13) return fail("bad scan parameters");
      if (!is_progressive && !(s.Ss == 0 && s.Se == 63 && s.Al == 0 && s.Ah == 0)) return fail("spectral selection in a sequential JPEG");
      const int mode = s.Ah != 0 ? 2 : is_progressive ? 1 : 0;    // EncodeScan<kMode>: sequential, progressive first pass, refinement
      EobState eob_state;
      const size_t len = 6 + 2 * s.num_components;
      out->push_back(0xFF); out->push_back(0xDA); out->push_back((uint8_t)(len >> 8)); out->push_back((uint8_t)len); out->push_back((uint8_t)s.num_components);
      for (uint32_t i = 0; i < s.num_components; i++) {
        out->push_back((uint8_t)jd.components[s.components[i].comp_idx].id);
        out->push_back((uint8_t)((s.components[i].dc_tbl_idx << 4) | s.components[i].ac_tbl_idx));
      }
      out->push_back((uint8_t)s.Ss); out->push_back((uint8_t)s.Se); out->push_back((uint8_t)((s.Ah << 4) | s.Al));
      // ---- entropy-coded segment (EncodeScan, sequential mode).  Interleaved scans: an MCU holds v_samp x h_samp blocks of every scan
      // component and the MCU grid covers the padded image; a single-component scan walks that component's own blocks, one per MCU,
      // over ceil(size * samp / (8 * max_samp)) of them (the blocks that hold image data).
      BitWriter w; w.out = out;
      int last_dc[4] = {0, 0, 0, 0};
      const uint32_t restart_interval = seen_dri ? jd.restart_interval : 0;
      uint32_t restarts_to_go = restart_interval, next_restart = 0, block_scan_index = 0;
      size_t ezr_pos = 0, reset_pos = 0;
      const bool interleaved = s.num_components > 1;
      uint32_t scan_cols = mcu_cols, scan_rows = mcu_rows;
      if (!interleaved) {
        const JpegComponentInfo& c0 = jd.components[s.components[0].comp_idx];
        scan_cols = (width * c0.h_samp + 8 * max_h - 1) / (8 * max_h);
        scan_rows = (height * c0.v_samp + 8 * max_v - 1) / (8 * max_v);
      }
      for (uint32_t my = 0; my < scan_rows; my++) for (uint32_t mx = 0; mx < scan_cols; mx++) {
        if (restart_interval > 0 && restarts_to_go == 0) {
          eob_state.Flush(w);
          if (!w.Pad(jd, &pad_pos)) return fail("padding bits exhausted");
          out->push_back(0xFF); out->push_back((uint8_t)(0xD0 + next_restart));
          next_restart = (next_restart + 1) & 7;
          restarts_to_go = restart_interval;
          memset(last_dc, 0, sizeof(last_dc));
        }
        for (uint32_t i = 0; i < s.num_components; i++) {
          const JpegScanComponent& sc = s.components[i];
          const JpegComponentInfo& comp = jd.components[sc.comp_idx];
          const uint32_t nby = interleaved ? comp.v_samp : 1, nbx = interleaved ? comp.h_samp : 1, comp_bw = mcu_cols * comp.h_samp;
          for (uint32_t iy = 0; iy < nby; iy++) for (uint32_t ix = 0; ix < nbx; ix++) {
          const HuffTable& dct = dc_tab[sc.dc_tbl_idx & 3];
          const HuffTable& act = ac_tab[sc.ac_tbl_idx & 3];
          if ((s.Ss == 0 && s.Ah == 0 && !dct.init) || (s.Se > 0 && !act.init)) return fail("scan uses an undefined Huffman table");
          if (reset_pos < s.reset_points.size() && s.reset_points[reset_pos] == block_scan_index) { eob_state.Flush(w); reset_pos++; }   // the original file ended its EOB run here
          int num_zero_runs = 0;
          if (ezr_pos < s.extra_zero_runs.size() && s.extra_zero_runs[ezr_pos].first == block_scan_index) num_zero_runs = (int)s.extra_zero_runs[ezr_pos++].second;
          const int16_t* c = coeffs[sc.comp_idx] + ((size_t)(my * nby + iy) * comp_bw + (mx * nbx + ix)) * 64;
          if (mode != 0) {
            const bool ok = mode == 1 ? EncodeBlockProgressive(c, dct, act, (int)s.Ss, (int)s.Se, (int)s.Al, num_zero_runs, eob_state, &last_dc[sc.comp_idx], w)
                                      : EncodeBlockRefinement(c, act, (int)s.Ss, (int)s.Se, (int)s.Al, eob_state, w);
            if (!ok) return fail("coefficient out of range");
            if (!w.ok) return fail("symbol without a Huffman code");
            block_scan_index++;
            continue;
          }
          // EncodeDCTBlockSequential
          int temp2 = c[0], temp = temp2 - last_dc[sc.comp_idx];
          last_dc[sc.comp_idx] = temp2;
          temp2 = temp;
          if (temp < 0) { temp = -temp; temp2--; }
          int dc_nbits = 0;
          while ((temp >> dc_nbits) != 0) dc_nbits++;
          if (dc_nbits >= 12) return fail("DC difference out of range");
          w.Symbol(dc_nbits, dct);
          if (dc_nbits > 0) w.Put((uint32_t)temp2 & ((1u << dc_nbits) - 1), dc_nbits);
          int r = 0;
          for (int k = 1; k < 64; k++) {
            temp = c[kNaturalOrder[k]];
            if (temp == 0) { r++; continue; }
            if (temp < 0) { temp = -temp; temp2 = ~temp; } else temp2 = temp;
            while (r > 15) { w.Symbol(0xF0, act); r -= 16; }
            int ac_nbits = 0;
            while ((temp >> ac_nbits) != 0) ac_nbits++;
            if (ac_nbits >= 16) return fail("AC coefficient out of range");
            w.Symbol((r << 4) + ac_nbits, act);
            w.Put((uint32_t)temp2 & ((1u << ac_nbits) - 1), ac_nbits);
            r = 0;
          }
          for (int k = 0; k < num_zero_runs; k++) { w.Symbol(0xF0, act); r -= 16; }
          if (r > 0) w.Symbol(0, act);
          if (!w.ok) return fail("symbol without a Huffman code");
          block_scan_index++;
          }
        }
        if (restart_interval > 0) restarts_to_go--;
      }
      eob_state.Flush(w);
      if (!w.ok) return fail("symbol without a Huffman code");
      if (!w.Pad(jd, &pad_pos)) return fail("padding bits exhausted");
    } else if (m == 0xD9) {
      out->push_back(0xFF); out->push_back(0xD9);
      out->insert(out->end(), jd.tail_data.begin(), jd.tail_data.end());
    } else return fail("unsupported marker");
  }
  return true;
}

}  // namespace jxlhip
