// jxl-hip: the one exchange step of the multi-GPU path behind the C ABI — the gather of decoded pixels to the consumer rank over RCCL / xGMI
// (BASELINE.json north_star: "RCCL over xGMI only for the gather of decoded pixels").  A Rust caller does not need PyTorch for it:
// one process per GPU, rank 0 makes a communicator id (JxlHipCommGetUniqueId) and hands it to the others by whatever means the job has
// (a file, an environment variable, MPI ...), every rank creates its communicator and calls JxlHipGatherFrames after its decode.
//
// xGMI is point to point (seven links per GPU): every peer's chunk of `chunk_frames` frames travels over its own link at the same time, grouped into one
// ncclGroupStart / ncclGroupEnd per chunk, and lands at its final position in the consumer's job buffer [world][frames][frame_bytes]; the consumer's own
// shard is a device copy.  librccl.so is loaded on first use (a single-GPU user of libjxl.so never needs it).
#include "../../include/jxl_hip.h"
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <algorithm>
#include <vector>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

extern "C" const char* JxlHipLastError(void);
namespace jxlhip { void SetLastErrorText(const std::string& s); }

namespace {
struct Rccl {
  // (nccl.h: ncclResult_t is an int, 0 = success; ncclUniqueId is 128 bytes; ncclDataType_t ncclUint8 = 1, ncclInt64 = 4; ncclRedOp_t ncclSum = 0)
  int (*GetUniqueId)(void* id) = nullptr;
  int (*CommInitRank)(void** comm, int nranks, const void* id_by_value_128, int rank) = nullptr;
  int (*CommDestroy)(void* comm) = nullptr;
  int (*Send)(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) = nullptr;
  int (*Recv)(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  int (*AllReduce)(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};
struct UniqueId { char bytes[128]; };
// ncclCommInitRank takes the id BY VALUE (a 128-byte struct): the typed pointer below has the right calling convention
typedef int (*CommInitRankFn)(void** comm, int nranks, UniqueId id, int rank);

Rccl& Lib() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    // JXL_HIP_RCCL_LIB: the library to bind instead (a test double that runs several ranks on one GPU: tests/fake_rccl)
    const char* named = getenv("JXL_HIP_RCCL_LIB");
    void* h = named && *named ? dlopen(named, RTLD_NOW | RTLD_LOCAL) : nullptr;
    if (named && *named && !h) return;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
#define SYM(field, name) *(void**)(&r.field) = dlsym(h, name)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy"); SYM(Send, "ncclSend"); SYM(Recv, "ncclRecv");
    SYM(GroupStart, "ncclGroupStart"); SYM(GroupEnd, "ncclGroupEnd"); SYM(AllReduce, "ncclAllReduce"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    r.ok = r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.Send && r.Recv && r.GroupStart && r.GroupEnd && r.AllReduce;
  });
  return r;
}
bool Fail(const std::string& what, int rc = 0) {
  Rccl& L = Lib();
  jxlhip::SetLastErrorText(what + (rc && L.GetErrorString ? std::string(": ") + L.GetErrorString(rc) : std::string()));
  return false;
}
}  // namespace

struct JxlHipCommStruct { void* comm = nullptr; int rank = 0, world = 1, device = 0; };

extern "C" {

int JxlHipCommGetUniqueId(uint8_t id[JXL_HIP_COMM_ID_BYTES]) {
  Rccl& L = Lib();
  if (!L.ok) { Fail("librccl.so could not be loaded"); return 1; }
  const int rc = L.GetUniqueId(id);
  if (rc) { Fail("ncclGetUniqueId", rc); return 1; }
  return 0;
}

JxlHipComm* JxlHipCommCreate(int device, int rank, int world, const uint8_t id[JXL_HIP_COMM_ID_BYTES]) {
  if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) { Fail("JxlHipCommCreate: bad rank / world / id"); return nullptr; }    // (a single rank needs no id exchange and no RCCL: jxl_hip.h)
  if (hipSetDevice(device) != hipSuccess) { (void)hipGetLastError(); Fail("JxlHipCommCreate: no such HIP device"); return nullptr; }
  JxlHipComm* c = new JxlHipCommStruct();
  c->rank = rank; c->world = world; c->device = device;
  if (world > 1) {
    Rccl& L = Lib();
    if (!L.ok) { delete c; Fail("librccl.so could not be loaded"); return nullptr; }
    UniqueId uid;
    memcpy(uid.bytes, id, sizeof uid.bytes);
    const int rc = reinterpret_cast<CommInitRankFn>(L.CommInitRank)(&c->comm, world, uid, rank);
    if (rc) { delete c; Fail("ncclCommInitRank", rc); return nullptr; }
  }
  return c;
}

void JxlHipCommDestroy(JxlHipComm* c) {
  if (!c) return;
  if (c->comm) (void)Lib().CommDestroy(c->comm);
  delete c;
}

int JxlHipGatherFramesRagged(JxlHipComm* c, const void* send, size_t frame_bytes, const int* frames_per_rank, void* recv, int root, int chunk_frames, void* hip_stream) {
  if (!c || !frames_per_rank || root < 0 || root >= c->world || frame_bytes == 0) { Fail("JxlHipGatherFrames: bad arguments"); return 1; }
  if (c->world > 1024) { Fail("JxlHipGatherFrames: more than 1024 ranks"); return 1; }
  hipStream_t s = (hipStream_t)hip_stream;
  const int mine = frames_per_rank[c->rank];
  if (mine < 0 || (mine > 0 && !send) || (c->rank == root && !recv)) { Fail("JxlHipGatherFrames: missing buffer"); return 1; }
  if (hipSetDevice(c->device) != hipSuccess) { (void)hipGetLastError(); Fail("JxlHipGatherFrames: no such HIP device"); return 1; }
  std::vector<size_t> off((size_t)c->world + 1);      // (first frame of every rank's shard in the consumer's buffer; off the stack: ADVICE r5)
  int longest = 0;
  off[0] = 0;
  for (int r = 0; r < c->world; r++) { if (frames_per_rank[r] < 0) { Fail("JxlHipGatherFrames: negative shard length"); return 1; } off[r + 1] = off[r] + (size_t)frames_per_rank[r]; longest = std::max(longest, frames_per_rank[r]); }
  uint8_t* out = (uint8_t*)recv;
  if (c->rank == root && mine > 0 && out + off[root] * frame_bytes != send)
    if (hipMemcpyAsync(out + off[root] * frame_bytes, send, (size_t)mine * frame_bytes, hipMemcpyDeviceToDevice, s) != hipSuccess) { (void)hipGetLastError(); Fail("JxlHipGatherFrames: device copy of the consumer's own shard failed"); return 1; }
  if (c->world == 1) return 0;
  Rccl& L = Lib();
  chunk_frames = std::max(1, chunk_frames);
  for (int c0 = 0; c0 < longest; c0 += chunk_frames) {
    int rc = L.GroupStart();
    if (rc) { Fail("ncclGroupStart", rc); return 1; }
    if (c->rank == root) {
      for (int r = 0; r < c->world && !rc; r++) {
        const int c1 = std::min(frames_per_rank[r], c0 + chunk_frames);
        if (r == root || c1 <= c0) continue;
        rc = L.Recv(out + (off[r] + (size_t)c0) * frame_bytes, (size_t)(c1 - c0) * frame_bytes, /*ncclUint8*/ 1, r, c->comm, s);
      }
    } else {
      const int c1 = std::min(mine, c0 + chunk_frames);
      if (c1 > c0) rc = L.Send((const uint8_t*)send + (size_t)c0 * frame_bytes, (size_t)(c1 - c0) * frame_bytes, 1, root, c->comm, s);
    }
    const int rc2 = L.GroupEnd();
    if (rc || rc2) { Fail("ncclSend / ncclRecv", rc ? rc : rc2); return 1; }
  }
  return 0;
}

int JxlHipGatherFrames(JxlHipComm* c, const void* send, size_t frame_bytes, int frames, void* recv, int root, int chunk_frames, void* hip_stream) {
  if (!c || frames < 0 || c->world > 1024) { Fail("JxlHipGatherFrames: bad arguments"); return 1; }
  std::vector<int> per((size_t)c->world, frames);
  return JxlHipGatherFramesRagged(c, send, frame_bytes, per.data(), recv, root, chunk_frames, hip_stream);
}

int JxlHipAllReduceSumI64(JxlHipComm* c, int64_t* device_values, size_t count, void* hip_stream) {
  if (!c || !device_values) { Fail("JxlHipAllReduceSumI64: bad arguments"); return 1; }
  if (c->world == 1) return 0;
  const int rc = Lib().AllReduce(device_values, device_values, count, /*ncclInt64*/ 4, /*ncclSum*/ 0, c->comm, (hipStream_t)hip_stream);
  if (rc) { Fail("ncclAllReduce", rc); return 1; }
  return 0;
}

}  // extern "C"
