// fuzz_host — mutation fuzzing of everything this library parses on the HOST, in a build whose host sources are compiled with
// -fsanitize=address,undefined (jpegxl-rs_amd/Makefile target `asan`).  The reference's CI runs its tests under ASan / TSan
// (/root/reference/.github/workflows/ci.yml:71-106); the arithmetic it sanitises there lives in libjxl — here the host half of it is ours:
// container + box walk, image / frame headers, TOC, LfGlobal / HfGlobal tables (MA trees, histograms -> alias tables, quantisation tables,
// coefficient orders), splines / patches / noise syntax, ICC stream, the jbrd box + JPEG marker rebuild (csrc/host_parse.cc, host_features.cc,
// icc_profile.cc, jpeg_recon.cc, the box scanner and state machine of jxl_abi.cc).  No GPU is needed: JxlHipDebugDescribe / JxlHipImageOutSize /
// JxlHipColorProfileFromHeaders are host-only, and the JxlDecoder state machine is driven up to the point where it asks for a device.
//
//   fuzz_host <corpus dir> <trials> <seconds> [seed]
// Every trial takes a corpus file, applies 1..8 mutations (bit flips, byte stores, truncation, splices, length-field edits) and feeds it through the
// entry points.  A sanitizer report aborts the process (non-zero exit); the summary line says how many inputs were accepted / rejected.
#include <dirent.h>
#include <signal.h>
#include <unistd.h>
#include <fcntl.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>
#include "../../include/jxl_hip.h"

static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
static uint64_t Rng() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return g_rng; }

// a trial that runs longer than 20 s is a finding, too (an input that makes the parser spin): the input is written to slow_input.jxl and the process fails
extern "C" void __sanitizer_print_stack_trace(void);      // (where the trial was when its time ran out)
static const uint8_t* g_cur = nullptr; static size_t g_cur_size = 0;
static void OnAlarm(int) {
  __sanitizer_print_stack_trace();
  const int fd = open("slow_input.jxl", O_WRONLY | O_CREAT | O_TRUNC, 0644);
  if (fd >= 0) { size_t off = 0; while (off < g_cur_size) { const ssize_t k = write(fd, g_cur + off, g_cur_size - off); if (k <= 0) break; off += (size_t)k; } close(fd); }
  static const char msg[] = "fuzz_host: a trial exceeded its time budget (input saved as slow_input.jxl)\n";
  (void)!write(2, msg, sizeof msg - 1);
  _exit(3);
}

static void Mutate(std::vector<uint8_t>& b, const std::vector<std::vector<uint8_t>>& corpus) {
  const int n = 1 + (int)(Rng() % 8);
  for (int k = 0; k < n && !b.empty(); k++) {
    const size_t pos = (Rng() % 4 == 0) ? Rng() % std::min<size_t>(b.size(), 96) : Rng() % b.size();   // headers get a quarter of the hits
    switch (Rng() % 8) {
      case 0: case 1: case 2: b[pos] ^= (uint8_t)(1u << (Rng() % 8)); break;
      case 3: b[pos] = (uint8_t)Rng(); break;
      case 4: b[pos] = (Rng() & 1) ? 0xFF : 0x00; break;
      case 5: b.resize(1 + Rng() % b.size()); break;                                                     // truncation
      case 6: {                                                                                         // splice a run of another file in
        const auto& o = corpus[Rng() % corpus.size()];
        if (o.empty()) break;
        const size_t len = 1 + Rng() % std::min<size_t>(o.size(), 64), src = Rng() % (o.size() - len + 1);
        for (size_t i = 0; i < len && pos + i < b.size(); i++) b[pos + i] = o[src + i];
        break;
      }
      default: {                                                                                        // a big-endian length field (boxes, jbrd markers)
        if (pos + 4 <= b.size()) { const uint32_t v = (uint32_t)Rng() >> (Rng() % 32); b[pos] = v >> 24; b[pos + 1] = v >> 16; b[pos + 2] = v >> 8; b[pos + 3] = v; }
        break;
      }
    }
  }
}

int main(int argc, char** argv) {
  if (argc < 4) { fprintf(stderr, "usage: %s <corpus dir> <trials> <seconds> [seed]\n", argv[0]); return 2; }
  std::vector<std::vector<uint8_t>> corpus;
  if (DIR* d = opendir(argv[1])) {
    std::vector<std::string> names;
    while (dirent* e = readdir(d)) if (e->d_name[0] != '.') names.push_back(e->d_name);
    closedir(d);
    std::sort(names.begin(), names.end());
    for (auto& n : names) {
      FILE* f = fopen((std::string(argv[1]) + "/" + n).c_str(), "rb");
      if (!f) continue;
      std::vector<uint8_t> b; uint8_t buf[65536]; size_t k;
      while ((k = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k);
      fclose(f);
      if (!b.empty() && b.size() <= (8u << 20)) corpus.push_back(std::move(b));
    }
  }
  if (corpus.empty()) { fprintf(stderr, "empty corpus\n"); return 2; }
  const long trials = atol(argv[2]);
  const double seconds = atof(argv[3]);
  if (argc > 4) g_rng ^= strtoull(argv[4], nullptr, 0) * 0xD1342543DE82EF95ull + 1;
  signal(SIGALRM, OnAlarm);
  const auto t0 = std::chrono::steady_clock::now();
  long done = 0, accepted = 0, rejected = 0, sized = 0, icc_ok = 0;
  std::vector<char> text(1 << 16);
  std::vector<uint8_t> icc(1 << 20);
  for (; done < trials; done++) {
    if ((done & 63) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > seconds) break;
    std::vector<uint8_t> b = corpus[(size_t)(done % (long)corpus.size())];
    if (done >= (long)corpus.size()) Mutate(b, corpus);            // (the first pass feeds the corpus as it is: everything must be accepted or rejected cleanly)
    // exact-size heap copy: an over-read of the input is an ASan report, not a read of vector slack
    uint8_t* data = (uint8_t*)malloc(b.size() ? b.size() : 1);
    memcpy(data, b.data(), b.size());
    g_cur = data; g_cur_size = b.size();
    alarm(20);
    (void)JxlSignatureCheck(data, b.size());
    if (JxlHipDebugDescribe(data, b.size(), text.data(), text.size()) == 0) accepted++; else rejected++;
    JxlPixelFormat fmt = {(uint32_t)(Rng() % 5), (Rng() & 1) ? JXL_TYPE_UINT8 : JXL_TYPE_FLOAT, JXL_NATIVE_ENDIAN, (size_t)(Rng() % 3 == 0 ? 16 : 0)};
    JxlBasicInfo info; size_t out_size = 0;
    if (JxlHipImageOutSize(data, b.size(), &fmt, &info, &out_size) == JXL_DEC_SUCCESS) sized++;
    size_t icc_size = icc.size();
    if (JxlHipColorProfileFromHeaders(data, b.size(), icc.data(), &icc_size) == 0) icc_ok++;
    // the decoder's state machine as far as it goes without a device (jpegxl-rs/src/decode.rs:207-325: create, subscribe, set input, close, process)
    if (JxlDecoder* dec = JxlDecoderCreate(nullptr)) {
      (void)JxlDecoderSubscribeEvents(dec, JXL_DEC_BASIC_INFO | JXL_DEC_COLOR_ENCODING | JXL_DEC_FULL_IMAGE | JXL_DEC_BOX | JXL_DEC_JPEG_RECONSTRUCTION);
      if (JxlDecoderSetInput(dec, data, b.size()) == JXL_DEC_SUCCESS) {
        if (Rng() & 1) JxlDecoderCloseInput(dec);
        for (int k = 0; k < 4; k++) { const JxlDecoderStatus st = JxlDecoderProcessInput(dec); if (st == JXL_DEC_ERROR || st == JXL_DEC_SUCCESS || st == JXL_DEC_NEED_MORE_INPUT) break; }
        (void)JxlDecoderGetBasicInfo(dec, &info);
      }
      JxlDecoderDestroy(dec);
    }
    alarm(0);
    free(data);
  }
  printf("{\"trials\": %ld, \"accepted\": %ld, \"rejected\": %ld, \"sized\": %ld, \"icc\": %ld, \"corpus\": %zu, \"seconds\": %.1f}\n", done, accepted, rejected, sized, icc_ok, corpus.size(),
         std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
  return 0;
}
