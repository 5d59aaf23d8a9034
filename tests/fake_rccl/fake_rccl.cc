// Test double of librccl for ONE GPU shared by several processes (tests/test_gather_abi.py): the entry points csrc/gather.cc binds — ncclGetUniqueId, ncclCommInitRank
// (the id BY VALUE, as nccl.h declares it), ncclSend / ncclRecv inside ncclGroupStart / ncclGroupEnd, ncclAllReduce (int64 sum), ncclCommDestroy, ncclGetErrorString —
// over a POSIX shared-memory segment named after the id: every ordered pair of ranks has a mailbox (header + 8 MiB of payload), a send stages device memory through
// host memory into it, a receive copies it out to the device; larger messages travel in pieces.  It exists so that the world > 1 control flow of JxlHipGatherFrames*
// (chunked groups, final positions in the consumer's buffer, ragged shards, the all-reduce) runs somewhere before an 8-GPU node runs it over xGMI; it says nothing about
// RCCL's performance.  JXL_HIP_RCCL_LIB names the library csrc/gather.cc loads.
#include <hip/hip_runtime.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <thread>
#include <vector>

namespace {
constexpr int kMaxRanks = 8;
constexpr size_t kSlotBytes = 8u << 20;
struct Mailbox { std::atomic<uint32_t> full; uint32_t pad; uint64_t bytes; };                   // followed by kSlotBytes of payload
struct Shared {
  std::atomic<uint32_t> arrived, generation;        // barrier
  char id[128];
  int64_t reduce[kMaxRanks][64];
};
struct Comm { Shared* sh; uint8_t* boxes; int rank, world; size_t map_bytes; char name[64]; };
struct Op { bool send; void* buf; size_t bytes; int peer; Comm* comm; hipStream_t stream; };
thread_local int g_group_depth = 0;
thread_local std::vector<Op>* g_ops = nullptr;

size_t BoxStride() { return sizeof(Mailbox) + kSlotBytes; }
Mailbox* Box(Comm* c, int src, int dst) { return reinterpret_cast<Mailbox*>(c->boxes + (size_t)(src * kMaxRanks + dst) * BoxStride()); }
bool WaitFor(std::atomic<uint32_t>& a, uint32_t want) {
  const auto t0 = std::chrono::steady_clock::now();
  while (a.load(std::memory_order_acquire) != want) {
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return false;
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  return true;
}
bool Barrier(Comm* c) {
  const uint32_t gen = c->sh->generation.load(std::memory_order_acquire);
  if (c->sh->arrived.fetch_add(1) + 1 == (uint32_t)c->world) { c->sh->arrived.store(0); c->sh->generation.fetch_add(1); return true; }
  return WaitFor(c->sh->generation, gen + 1);
}
size_t TypeBytes(int dtype) { return dtype == 1 /*ncclUint8*/ || dtype == 0 /*ncclInt8*/ ? 1 : dtype == 4 /*ncclInt64*/ || dtype == 5 || dtype == 8 ? 8 : 4; }

int RunOps(std::vector<Op>& ops) {
  // sends and receives of a group progress together (a rank both sends and receives in a ring-like pattern must not block on its first operation): round-robin over
  // the operations, each moving one piece of at most kSlotBytes whenever its mailbox is ready
  std::vector<size_t> done(ops.size(), 0);
  std::vector<uint8_t> host(kSlotBytes);
  size_t remaining = ops.size();
  for (auto& o : ops) if (o.bytes == 0) remaining--;
  const auto t0 = std::chrono::steady_clock::now();
  while (remaining) {
    bool progressed = false;
    for (size_t i = 0; i < ops.size(); i++) {
      Op& o = ops[i];
      if (done[i] >= o.bytes) continue;
      Mailbox* b = o.send ? Box(o.comm, o.comm->rank, o.peer) : Box(o.comm, o.peer, o.comm->rank);
      uint8_t* payload = reinterpret_cast<uint8_t*>(b + 1);
      if (o.send) {
        if (b->full.load(std::memory_order_acquire) != 0) continue;
        const size_t n = std::min(kSlotBytes, o.bytes - done[i]);
        if (hipMemcpyAsync(payload, (const uint8_t*)o.buf + done[i], n, hipMemcpyDeviceToHost, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) return 1;
        b->bytes = n;
        b->full.store(1, std::memory_order_release);
        done[i] += n;
      } else {
        if (b->full.load(std::memory_order_acquire) != 1) continue;
        const size_t n = (size_t)b->bytes;
        if (n > o.bytes - done[i]) return 5;                                    // (the peer sent more than this receive expects)
        memcpy(host.data(), payload, n);
        b->full.store(0, std::memory_order_release);
        if (hipMemcpyAsync((uint8_t*)o.buf + done[i], host.data(), n, hipMemcpyHostToDevice, o.stream) != hipSuccess || hipStreamSynchronize(o.stream) != hipSuccess) return 1;
        done[i] += n;
      }
      progressed = true;
      if (done[i] >= o.bytes) remaining--;
    }
    if (!progressed) {
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(60)) return 6;
      std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  }
  return 0;
}
}  // namespace

extern "C" {
typedef struct { char internal[128]; } ncclUniqueId;

int ncclGetUniqueId(ncclUniqueId* id) {
  memset(id->internal, 0, sizeof id->internal);
  static std::atomic<uint32_t> counter{0};
  snprintf(id->internal, sizeof id->internal, "fake_rccl_%d_%u_%llx", (int)getpid(), counter.fetch_add(1), (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  for (size_t i = strlen(id->internal) + 1; i < sizeof id->internal; i++) id->internal[i] = (char)(0x40 + i % 37);      // (every byte of the id has to arrive: the tail is checked below)
  return 0;
}

int ncclCommInitRank(void** comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > kMaxRanks || rank < 0 || rank >= nranks) return 4;
  if (strnlen(id.internal, sizeof id.internal) >= 60 || strncmp(id.internal, "fake_rccl_", 10) != 0) return 4;        // the id did not arrive by value as nccl.h passes it
  for (size_t i = strlen(id.internal) + 1; i < sizeof id.internal; i++) if (id.internal[i] != (char)(0x40 + i % 37)) return 4;
  Comm* c = new Comm();
  c->rank = rank; c->world = nranks;
  snprintf(c->name, sizeof c->name, "/%s", id.internal);
  c->map_bytes = sizeof(Shared) + (size_t)kMaxRanks * kMaxRanks * BoxStride();
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) { delete c; return 2; }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) { delete c; return 2; }
  c->sh = reinterpret_cast<Shared*>(p);                     // (a fresh segment is all zeros: counters and mailbox flags start at 0)
  c->boxes = reinterpret_cast<uint8_t*>(p) + sizeof(Shared);
  if (rank == 0) memcpy(c->sh->id, id.internal, sizeof id.internal);
  if (!Barrier(c)) { delete c; return 6; }
  if (memcmp(c->sh->id, id.internal, sizeof id.internal) != 0) { delete c; return 4; }
  *comm = c;
  return 0;
}

int ncclCommDestroy(void* comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return 0;
  Barrier(c);
  if (c->rank == 0) shm_unlink(c->name);
  munmap(c->sh, c->map_bytes);
  delete c;
  return 0;
}

int ncclGroupStart() { if (g_group_depth++ == 0) { delete g_ops; g_ops = new std::vector<Op>(); } return 0; }
int ncclGroupEnd() {
  if (g_group_depth <= 0) return 5;
  if (--g_group_depth) return 0;
  std::vector<Op> ops; ops.swap(*g_ops);
  return RunOps(ops);
}
static int Enqueue(bool send, void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || peer < 0 || peer >= c->world || peer == c->rank) return 4;
  Op o{send, buf, count * TypeBytes(dtype), peer, c, s};
  if (g_group_depth) { g_ops->push_back(o); return 0; }
  std::vector<Op> one{o};
  return RunOps(one);
}
int ncclSend(const void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) { return Enqueue(true, const_cast<void*>(buf), count, dtype, peer, comm, s); }
int ncclRecv(void* buf, size_t count, int dtype, int peer, void* comm, hipStream_t s) { return Enqueue(false, buf, count, dtype, peer, comm, s); }

int ncclAllReduce(const void* send, void* recv, size_t count, int dtype, int op, void* comm, hipStream_t s) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || dtype != 4 || op != 0 || count > 64) return 4;      // int64 sums of up to 64 values: what csrc/gather.cc asks for
  int64_t mine[64];
  if (hipMemcpyAsync(mine, send, count * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
  memcpy(c->sh->reduce[c->rank], mine, count * 8);
  if (!Barrier(c)) return 6;
  int64_t sum[64] = {0};
  for (int r = 0; r < c->world; r++) for (size_t i = 0; i < count; i++) sum[i] += c->sh->reduce[r][i];
  if (!Barrier(c)) return 6;
  if (hipMemcpyAsync(recv, sum, count * 8, hipMemcpyHostToDevice, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return 1;
  return 0;
}

const char* ncclGetErrorString(int rc) {
  switch (rc) { case 0: return "no error"; case 1: return "unhandled cuda error (fake)"; case 2: return "unhandled system error (fake)"; case 4: return "invalid argument (fake)"; case 5: return "invalid usage (fake)"; case 6: return "timeout (fake)"; default: return "error (fake)"; }
}
}  // extern "C"
