// api_concurrent — what the UNCHANGED Rust crate does with the libjxl C ABI, from T host threads at once, without a Rust toolchain.
//
// Every thread owns one JxlDecoder (decoders are Send: jpegxl-rs/src/decode.rs:523-532) and runs the event loop of decode_internal
// (jpegxl-rs/src/decode.rs:207-325) per file: SubscribeEvents(BASIC_INFO | FULL_IMAGE), SetInput, CloseInput, ProcessInput until Success; at
// NeedImageOutBuffer it asks for the size, zero-fills a fresh buffer (Vec::resize(size, 0), decode.rs:417-421) and registers it; Reset after Success.
// Host bytes in, host pixels out.  The library is loaded with dlopen (the path is an argument), so the same binary times any libjxl.so with that ABI —
// this repository's, or a real libjxl if a box has one (benches/decode.rs:16-37 shape).
//
//   api_concurrent <libjxl.so> <dir with *.jxl> <threads,comma separated> <decodes per thread> [channels=3] [verify]
// prints one JSON line per thread count; with `verify` also {"crc32": {file: crc of the pixels}} of every file (first decode), for the caller to
// compare with the oracle's.
#include <dlfcn.h>
#include <dirent.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <thread>
#include <vector>
#include "../include/jxl_hip.h"

namespace {
struct Api {
  JxlDecoder* (*Create)(const JxlMemoryManager*);
  void (*Reset)(JxlDecoder*);
  void (*Destroy)(JxlDecoder*);
  JxlDecoderStatus (*Subscribe)(JxlDecoder*, int);
  JxlDecoderStatus (*SetInput)(JxlDecoder*, const uint8_t*, size_t);
  void (*CloseInput)(JxlDecoder*);
  JxlDecoderStatus (*Process)(JxlDecoder*);
  JxlDecoderStatus (*GetBasicInfo)(const JxlDecoder*, JxlBasicInfo*);
  JxlDecoderStatus (*OutSize)(const JxlDecoder*, const JxlPixelFormat*, size_t*);
  JxlDecoderStatus (*SetOut)(JxlDecoder*, const JxlPixelFormat*, void*, size_t);
  const char* (*LastError)();
};
uint32_t Crc32(const uint8_t* p, size_t n) {
  static uint32_t table[8][256];
  static bool init = false;
  if (!init) {
    for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; table[0][i] = c; }
    for (uint32_t i = 0; i < 256; i++) for (int t = 1; t < 8; t++) table[t][i] = table[0][table[t - 1][i] & 255] ^ (table[t - 1][i] >> 8);
    init = true;
  }
  uint32_t c = 0xFFFFFFFFu;
  while (n >= 8) {
    uint32_t a, b; memcpy(&a, p, 4); memcpy(&b, p + 4, 4);
    a ^= c;
    c = table[7][a & 255] ^ table[6][(a >> 8) & 255] ^ table[5][(a >> 16) & 255] ^ table[4][a >> 24] ^ table[3][b & 255] ^ table[2][(b >> 8) & 255] ^ table[1][(b >> 16) & 255] ^ table[0][b >> 24];
    p += 8; n -= 8;
  }
  while (n--) c = table[0][(c ^ *p++) & 255] ^ (c >> 8);
  return ~c;
}
// one decode, the way decode.rs does it; returns the pixel bytes (empty = failed)
bool DecodeOne(const Api& api, JxlDecoder* dec, const std::vector<uint8_t>& file, uint32_t channels, std::vector<uint8_t>* pixels, JxlBasicInfo* info) {
  if (api.Subscribe(dec, JXL_DEC_BASIC_INFO | JXL_DEC_FULL_IMAGE) != JXL_DEC_SUCCESS) return false;
  if (api.SetInput(dec, file.data(), file.size()) != JXL_DEC_SUCCESS) return false;
  api.CloseInput(dec);
  for (;;) {
    const JxlDecoderStatus st = api.Process(dec);
    if (st == JXL_DEC_BASIC_INFO) { if (api.GetBasicInfo(dec, info) != JXL_DEC_SUCCESS) return false; }
    else if (st == JXL_DEC_NEED_IMAGE_OUT_BUFFER) {
      JxlPixelFormat f = {channels, JXL_TYPE_UINT8, JXL_NATIVE_ENDIAN, 0};
      size_t size = 0;
      if (api.OutSize(dec, &f, &size) != JXL_DEC_SUCCESS) return false;
      pixels->clear(); pixels->resize(size, 0);
      if (api.SetOut(dec, &f, pixels->data(), size) != JXL_DEC_SUCCESS) return false;
    } else if (st == JXL_DEC_FULL_IMAGE) continue;
    else if (st == JXL_DEC_SUCCESS) { api.Reset(dec); return true; }
    else return false;
  }
}
}  // namespace

int main(int argc, char** argv) {
  if (argc < 5) { fprintf(stderr, "usage: %s <libjxl.so> <dir> <threads,...> <decodes per thread> [channels] [verify]\n", argv[0]); return 2; }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
  if (!lib) { fprintf(stderr, "dlopen: %s\n", dlerror()); return 2; }
  Api api;
  memset(&api, 0, sizeof(api));
#define SYM(field, name) *(void**)(&api.field) = dlsym(lib, name); if (!api.field && strcmp(name, "JxlHipLastError")) { fprintf(stderr, "missing symbol %s\n", name); return 2; }
  SYM(Create, "JxlDecoderCreate") SYM(Reset, "JxlDecoderReset") SYM(Destroy, "JxlDecoderDestroy") SYM(Subscribe, "JxlDecoderSubscribeEvents") SYM(SetInput, "JxlDecoderSetInput")
  SYM(CloseInput, "JxlDecoderCloseInput") SYM(Process, "JxlDecoderProcessInput") SYM(GetBasicInfo, "JxlDecoderGetBasicInfo") SYM(OutSize, "JxlDecoderImageOutBufferSize")
  SYM(SetOut, "JxlDecoderSetImageOutBuffer") SYM(LastError, "JxlHipLastError")
  std::vector<std::string> names;
  if (DIR* d = opendir(argv[2])) {
    while (dirent* e = readdir(d)) { const std::string n = e->d_name; if (n.size() > 4 && n.substr(n.size() - 4) == ".jxl") names.push_back(n); }
    closedir(d);
  }
  std::sort(names.begin(), names.end());
  if (names.empty()) { fprintf(stderr, "no .jxl files in %s\n", argv[2]); return 2; }
  std::vector<std::vector<uint8_t>> files;
  for (auto& n : names) {
    const std::string path = std::string(argv[2]) + "/" + n;
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) { fprintf(stderr, "cannot read %s\n", path.c_str()); return 2; }
    std::vector<uint8_t> b; uint8_t buf[65536]; size_t k;
    while ((k = fread(buf, 1, sizeof buf, f)) > 0) b.insert(b.end(), buf, buf + k);
    fclose(f);
    files.push_back(std::move(b));
  }
  std::vector<int> thread_counts;
  for (char* tok = strtok(argv[3], ","); tok; tok = strtok(nullptr, ",")) thread_counts.push_back(std::max(1, atoi(tok)));
  const int per_thread = std::max(1, atoi(argv[4]));
  const uint32_t channels = argc > 5 ? (uint32_t)atoi(argv[5]) : 3;
  const bool verify = argc > 6 && !strcmp(argv[6], "verify");
  if (verify) {
    JxlDecoder* dec = api.Create(nullptr);
    printf("{\"crc32\": {");
    for (size_t i = 0; i < files.size(); i++) {
      std::vector<uint8_t> px; JxlBasicInfo info;
      if (!DecodeOne(api, dec, files[i], channels, &px, &info)) { fprintf(stderr, "decode of %s failed: %s\n", names[i].c_str(), api.LastError ? api.LastError() : ""); return 1; }
      printf("%s\"%s\": %u", i ? ", " : "", names[i].c_str(), Crc32(px.data(), px.size()));
    }
    printf("}}\n");
    api.Destroy(dec);
  }
  for (int T : thread_counts) {
    for (int round = 0; round < 2; round++) {      // round 0 = warm-up (device arenas, staging buffers, the scheduler's threads), round 1 is timed
      std::atomic<int> ready{0}, failures{0};
      std::atomic<bool> go{false};
      std::atomic<uint64_t> pixels{0};
      std::vector<double> latency((size_t)T * per_thread, 0.0);
      std::vector<std::thread> threads;
      for (int t = 0; t < T; t++) threads.emplace_back([&, t] {
        JxlDecoder* dec = api.Create(nullptr);
        std::vector<uint8_t> px; JxlBasicInfo info;
        ready++;
        while (!go.load()) std::this_thread::yield();
        for (int k = 0; k < per_thread; k++) {
          const auto t0 = std::chrono::steady_clock::now();
          const auto& f = files[((size_t)t + (size_t)k * T) % files.size()];
          if (!DecodeOne(api, dec, f, channels, &px, &info)) { failures++; continue; }
          pixels += (uint64_t)info.xsize * info.ysize;
          latency[(size_t)t * per_thread + k] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
        api.Destroy(dec);
      });
      while (ready.load() < T) std::this_thread::yield();
      const auto t0 = std::chrono::steady_clock::now();
      go = true;
      for (auto& th : threads) th.join();
      const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (round == 0) continue;
      std::sort(latency.begin(), latency.end());
      printf("{\"threads\": %d, \"decodes\": %d, \"failures\": %d, \"seconds\": %.4f, \"mpixel_per_s\": %.1f, \"latency_ms_median\": %.1f, \"latency_ms_p90\": %.1f}\n", T, T * per_thread, failures.load(), s,
             pixels.load() / 1e6 / s, latency[latency.size() / 2], latency[latency.size() * 9 / 10]);
      fflush(stdout);
    }
  }
  return 0;
}
