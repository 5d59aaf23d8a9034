// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into or called from the product path.
//
// CPU restatement of the JPEG XL bit-level syntax used by libjxl v0.11.2 (the un-vendored submodule the
// reference pins at jpegxl-sys/Cargo.toml:11 and jpegxl-sys/src/lib.rs:79).  libjxl's sources are absent
// from /root/reference, so every function cites the upstream file it restates plus the SURVEY.md App. B
// paragraph that records the fixture-verified ([V]) or recalled ([R]) form of the algorithm.
#pragma once
#include <cstdint>
#include <cstddef>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>
#include <cmath>

namespace jxlo {

struct Error : std::runtime_error {
  explicit Error(const std::string& s) : std::runtime_error(s) {}
};
#define JXLO_FAIL(msg) throw ::jxlo::Error(std::string(msg) + " @" + __FILE__ + ":" + std::to_string(__LINE__))
#define JXLO_CHECK(c) do { if (!(c)) JXLO_FAIL("check failed: " #c); } while (0)

// libjxl lib/jxl/dec_bit_reader.h : LSB-first bit reader (SURVEY App. B preamble [V]).
struct BitReader {
  const uint8_t* data = nullptr;
  size_t size = 0;      // bytes
  size_t pos = 0;       // bit position
  BitReader() {}
  BitReader(const uint8_t* d, size_t n) : data(d), size(n) {}
  inline uint64_t peek(int n) const {  // n <= 56
    uint64_t v = 0;
    size_t byte = pos >> 3;
    int sh = pos & 7;
    // gather up to 8 bytes (zero beyond the end, overrun is detected by checks on pos)
    for (int i = 0; i < 8; i++) {
      size_t b = byte + i;
      uint64_t x = b < size ? data[b] : 0;
      v |= x << (8 * i);
    }
    v >>= sh;
    return n == 0 ? 0 : (v & ((n >= 64) ? ~0ull : ((1ull << n) - 1)));
  }
  inline void skip(size_t n) { pos += n; }
  inline uint32_t u(int n) {
    if (n == 0) return 0;
    uint64_t v = peek(n);
    pos += n;
    if (pos > size * 8) JXLO_FAIL("bitstream overrun");
    return (uint32_t)v;
  }
  inline bool Bool() { return u(1) != 0; }
  void byte_align() { pos = (pos + 7) & ~size_t(7); }
  // zero-padding check is not enforced (libjxl does; not needed for parity)
  size_t bits_left() const { return size * 8 > pos ? size * 8 - pos : 0; }
};

// lib/jxl/fields.h U32 distributions (SURVEY App. B preamble).
struct U32Dist { int bits; uint32_t off; };
inline U32Dist Val(uint32_t v) { return {0, v}; }
inline U32Dist Bits(int n) { return {n, 0}; }
inline U32Dist BitsOffset(int n, uint32_t off) { return {n, off}; }
inline uint32_t U32(BitReader& br, U32Dist d0, U32Dist d1, U32Dist d2, U32Dist d3) {
  uint32_t sel = br.u(2);
  U32Dist d = sel == 0 ? d0 : sel == 1 ? d1 : sel == 2 ? d2 : d3;
  return d.off + br.u(d.bits);
}
// lib/jxl/fields.cc U64Coder::Read
inline uint64_t U64(BitReader& br) {
  uint32_t sel = br.u(2);
  if (sel == 0) return 0;
  if (sel == 1) return 1 + br.u(4);
  if (sel == 2) return 17 + br.u(8);
  uint64_t v = br.u(12);
  int shift = 12;
  while (br.u(1)) {
    if (shift == 60) { v |= (uint64_t)br.u(4) << shift; break; }
    v |= (uint64_t)br.u(8) << shift;
    shift += 8;
  }
  return v;
}
// lib/jxl/fields.cc F16Coder::Read (IEEE half → float; inf/nan rejected)
inline float F16(BitReader& br) {
  uint32_t b = br.u(16);
  uint32_t sign = b >> 15, exp = (b >> 10) & 31, mant = b & 1023;
  if (exp == 31) JXLO_FAIL("F16 inf/nan");
  float v;
  if (exp == 0) v = std::ldexp((float)mant, -24);
  else v = std::ldexp((float)(mant + 1024), (int)exp - 25);
  return sign ? -v : v;
}
inline uint32_t Enum(BitReader& br) { return U32(br, Val(0), Val(1), BitsOffset(4, 2), BitsOffset(6, 18)); }
inline int32_t UnpackSigned(uint32_t u) { return (int32_t)((u >> 1) ^ (~(u & 1) + 1)); }
inline int CeilLog2(uint32_t x) {  // ceil(log2(x)), x>=1
  int r = 0;
  while ((1ull << r) < x) r++;
  return r;
}
inline int FloorLog2(uint32_t x) {  // x>=1
  int r = 0;
  while (x >>= 1) r++;
  return r;
}
// extensions: U64 bitmask + per-bit U64 length, payload skipped (fields.cc BeginExtensions/EndExtensions)
inline void SkipExtensions(BitReader& br) {
  uint64_t ext = U64(br);
  if (!ext) return;
  uint64_t total = 0;
  for (int i = 0; i < 64; i++) if (ext >> i & 1) total += U64(br);
  br.skip(total);
}

}  // namespace jxlo
