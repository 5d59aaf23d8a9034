// jxl-hip: libjxl_threads look-alike — the nine symbols jpegxl-rs binds
// (jpegxl-sys/src/threads/thread_parallel_runner.rs:44-65, resizable_parallel_runner.rs:42-67; used at
// jpegxl-rs/src/parallel/threads_runner.rs:44-87 and resizable_runner.rs:40-85).  A plain fork-join pool: the GPU
// decode path does not need it, but callers may hand the runner to other libjxl-style consumers, so it must work.
#include "../../include/jxl_hip.h"
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

namespace {

struct Pool {
  JxlMemoryManager mm; bool has_mm;
  std::vector<std::thread>* threads;   // heap-allocated lazily so that the object itself stays tiny (< 1 KiB from the
  std::mutex* mu;                      // caller's allocator: threads_runner.rs:97-102)
  std::condition_variable* cv_work; std::condition_variable* cv_done;
  size_t num_workers;
  // job
  JxlParallelRunFunction func; void* opaque; std::atomic<uint32_t>* next; uint32_t end;
  uint64_t generation; size_t active; bool stop;
};

void Worker(Pool* p, size_t id, uint64_t seen) {   // seen: the pool's job generation when the worker was created — jobs that ran before
  std::unique_lock<std::mutex> lk(*p->mu);          // it existed are not its business (SetThreads restarts workers), later ones are
  for (;;) {
    p->cv_work->wait(lk, [&] { return p->stop || p->generation != seen; });
    if (p->stop) return;
    seen = p->generation;
    JxlParallelRunFunction func = p->func; void* opaque = p->opaque; uint32_t end = p->end;
    lk.unlock();
    for (;;) { uint32_t i = p->next->fetch_add(1); if (i >= end) break; func(opaque, i, id); }
    lk.lock();
    if (--p->active == 0) p->cv_done->notify_all();
  }
}

void StartWorkers(Pool* p, size_t n) {
  p->num_workers = n;
  uint64_t gen;
  { std::lock_guard<std::mutex> g(*p->mu); gen = p->generation; }
  for (size_t i = 0; i < n; i++) p->threads->emplace_back(Worker, p, i, gen);
}
void StopWorkers(Pool* p) {
  { std::lock_guard<std::mutex> g(*p->mu); p->stop = true; }
  p->cv_work->notify_all();
  for (auto& t : *p->threads) t.join();
  p->threads->clear();
  p->stop = false;
}

Pool* CreatePool(const JxlMemoryManager* mm, size_t workers) {
  JxlMemoryManager copy = {nullptr, nullptr, nullptr};
  bool has = false;
  if (mm) { if (!!mm->alloc != !!mm->free) return nullptr; if (mm->alloc) { copy = *mm; has = true; } }
  void* mem = has ? copy.alloc(copy.opaque, sizeof(Pool)) : malloc(sizeof(Pool));
  if (!mem) return nullptr;
  Pool* p = new (mem) Pool();
  p->mm = copy; p->has_mm = has;
  p->threads = new std::vector<std::thread>(); p->mu = new std::mutex(); p->cv_work = new std::condition_variable(); p->cv_done = new std::condition_variable();
  p->next = new std::atomic<uint32_t>(0);
  p->generation = 0; p->active = 0; p->stop = false; p->func = nullptr; p->opaque = nullptr; p->end = 0;
  StartWorkers(p, workers);
  return p;
}
void DestroyPool(Pool* p) {
  if (!p) return;
  StopWorkers(p);
  delete p->threads; delete p->mu; delete p->cv_work; delete p->cv_done; delete p->next;
  JxlMemoryManager mm = p->mm; bool has = p->has_mm;
  p->~Pool();
  if (has) mm.free(mm.opaque, p); else free(p);
}

int Run(Pool* p, void* jpegxl_opaque, JxlParallelRunInit init, JxlParallelRunFunction func, uint32_t start, uint32_t end) {
  if (start > end) return -1;
  if (start == end) return 0;
  int r = init(jpegxl_opaque, p->num_workers ? p->num_workers : 1);
  if (r != 0) return r;
  if (p->num_workers == 0) { for (uint32_t i = start; i < end; i++) func(jpegxl_opaque, i, 0); return 0; }
  std::unique_lock<std::mutex> lk(*p->mu);
  p->func = func; p->opaque = jpegxl_opaque; p->next->store(start); p->end = end;
  p->active = p->num_workers; p->generation++;
  p->cv_work->notify_all();
  p->cv_done->wait(lk, [&] { return p->active == 0; });
  p->func = nullptr; p->opaque = nullptr; p->end = 0;   // nothing stale for a later worker to pick up
  return 0;
}

}  // namespace

extern "C" {
JxlParallelRetCode JxlThreadParallelRunner(void* ro, void* jo, JxlParallelRunInit init, JxlParallelRunFunction func, uint32_t s, uint32_t e) { return Run((Pool*)ro, jo, init, func, s, e); }
void* JxlThreadParallelRunnerCreate(const JxlMemoryManager* mm, size_t n) { return CreatePool(mm, n); }
void JxlThreadParallelRunnerDestroy(void* ro) { DestroyPool((Pool*)ro); }
size_t JxlThreadParallelRunnerDefaultNumWorkerThreads(void) { unsigned n = std::thread::hardware_concurrency(); return n ? n : 1; }
JxlParallelRetCode JxlResizableParallelRunner(void* ro, void* jo, JxlParallelRunInit init, JxlParallelRunFunction func, uint32_t s, uint32_t e) { return Run((Pool*)ro, jo, init, func, s, e); }
void* JxlResizableParallelRunnerCreate(const JxlMemoryManager* mm) { return CreatePool(mm, 0); }
void JxlResizableParallelRunnerSetThreads(void* ro, size_t n) { Pool* p = (Pool*)ro; StopWorkers(p); StartWorkers(p, n); }
uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize) {
  // libjxl: one thread per ~(xsize*ysize / 2^17.. ) capped by hardware; mirror the documented intent: scale with megapixels
  uint64_t n = xsize * ysize / (1u << 16);
  unsigned hw = std::thread::hardware_concurrency();
  if (n < 1) n = 1;
  if (hw && n > hw) n = hw;
  return (uint32_t)n;
}
void JxlResizableParallelRunnerDestroy(void* ro) { DestroyPool((Pool*)ro); }
}
