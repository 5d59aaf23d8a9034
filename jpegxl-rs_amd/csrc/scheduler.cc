// jxl-hip: one shared decode pipeline per device behind the libjxl-compatible API (pipeline.h: SchedulerDecode).
//
// jpegxl-rs callers decode one image per JxlDecoderProcessInput call, and many of them run at once on many threads — decoders are Send
// (jpegxl-rs/src/decode.rs:523-532).  One image at a time leaves the GPU idle: a 4K frame is four serial LF chains and 135 group streams.  The
// scheduler turns whatever requests are waiting into one job of the device's pipeline (requests that arrive while a job is being prepared ride in
// the next one), jobs overlap in the pipeline, decoded pixels come back through pinned staging buffers and every caller copies its own image
// into its own buffer.  A single caller sees the latency of a one-image job on streams of the pipeline's own (never the NULL stream).
#include "pipeline.h"
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace jxlhip {
namespace {

struct Request {
  const uint8_t* data; size_t size; OutputSpec spec; void* dst; size_t dst_size;
  void* staging = nullptr; size_t staging_cap = 0;
  int status = -1; std::string error;
  bool done = false, retried = false;
  std::condition_variable cv;
};

bool SameSpec(const OutputSpec& a, const OutputSpec& b) {
  return a.num_channels == b.num_channels && a.type == b.type && a.big_endian == b.big_endian && a.align == b.align && a.keep_orientation == b.keep_orientation &&
         a.unpremul_alpha == b.unpremul_alpha && a.upto_frame == b.upto_frame && a.only_frame == b.only_frame && a.alpha_from_extra == b.alpha_from_extra &&
         a.int_bits == b.int_bits && a.render_spotcolors == b.render_spotcolors;
}

std::atomic<int> g_live_decoders{0};      // JxlDecoderCreate / Destroy (SchedulerNoteDecoder)
int EnvInt(const char* name, int def) { const char* e = getenv(name); return e && *e ? atoi(e) : def; }
double NowMs() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
bool Trace() { static const bool on = getenv("JXL_HIP_SCHED_TRACE") != nullptr; return on; }

class DeviceScheduler {
 public:
  explicit DeviceScheduler(int device) : device_(device) {
    PipelineOptions o;
    // latency mode: a handful of small jobs in flight; small jobs take the one-wavefront-per-stream LF kernel (a third of the SIMT kernel's latency on a GPU that has room)
    o.in_flight = EnvInt("JXL_HIP_SCHED_IN_FLIGHT", 4);      // (few slots: their arenas settle after a handful of jobs — hipMalloc under a busy GPU takes hundreds of milliseconds)
    o.lf_streams = EnvInt("JXL_HIP_SCHED_LF_STREAMS", 4);     // (with the HF, main, copy and upload streams: 13 of the 16 hardware queues — streams that share one serialise)
    // (the HF stage of a small job is a latency chain of its own, ~35 ms for one 4K frame: several in flight, a coefficient set each)
    o.hf_streams = EnvInt("JXL_HIP_SCHED_HF_STREAMS", 3);
    o.no_flag_wait = 1;
    o.hf_sparse = EnvInt("JXL_HIP_SCHED_HF_SPARSE", 1);
    o.prepare_threads = EnvInt("JXL_HIP_SCHED_PREPARE_THREADS", 3);
    o.parse_threads = EnvInt("JXL_HIP_SCHED_PARSE_THREADS", 8);
    // latency over occupancy: every LF-group stream gets a wavefront of its own (100 ms per launch instead of the 250-300 ms of the SIMT form, which packs eight streams
    // into a wavefront for the deep throughput pipeline)
    o.lane_stride_lf = EnvInt("JXL_HIP_SCHED_LANE_STRIDE_LF", 64);
    o.timed = Trace() ? 1 : 0;
    o.wide_first = 0;
    o.small_job_frames = EnvInt("JXL_HIP_SCHED_WIDE_BELOW", 0);
    max_job_ = std::max(1, EnvInt("JXL_HIP_SCHED_MAX_JOB", 64)); max_job_fixed_ = getenv("JXL_HIP_SCHED_MAX_JOB") != nullptr;
    // shared planes for a few 4K frames to start with (a process that decodes one picture must not pay for gigabytes); they grow to the largest job seen whenever the pipeline
    // idles, jobs that do not fit run on arenas of their own
    o.reserve_frames = EnvInt("JXL_HIP_SCHED_RESERVE_FRAMES", 4); o.reserve_width = 3840; o.reserve_height = 2160;
    max_jobs_ = std::max(1, EnvInt("JXL_HIP_SCHED_JOBS", 4));
    return_us_ = std::max(0, EnvInt("JXL_HIP_SCHED_RETURN_US", 25000));
    quiet_us_ = std::max(0, EnvInt("JXL_HIP_SCHED_QUIET_US", 1000)); max_wait_us_ = std::max(0, EnvInt("JXL_HIP_SCHED_MAX_WAIT_US", 8000));
    pipe_.reset(new Pipeline(device, o));
    collector_ = std::thread([this] { CollectorLoop(); });
    completer_ = std::thread([this] { CompleterLoop(); });
  }
  // Shutdown order: the collector first — it may be inside SubmitJob and push one more job —, then the completer, which only leaves once the collector has and
  // nothing is in flight (every request gets its Finish), then the callers that are still inside Decode (copying their pixels out of the staging buffers freed below).
  ~DeviceScheduler() {
    { std::lock_guard<std::mutex> lock(mu_); shutdown_ = true; }
    cv_.notify_all(); done_cv_.notify_all();
    if (collector_.joinable()) collector_.join();
    { std::lock_guard<std::mutex> lock(mu_); collector_done_ = true; }
    done_cv_.notify_all();
    if (completer_.joinable()) completer_.join();
    { std::unique_lock<std::mutex> lock(mu_); idle_cv_.wait(lock, [&] { return callers_ == 0; }); }
    pipe_.reset();
    (void)hipSetDevice(device_);
    for (auto& b : free_staging_) (void)hipHostFree(b.first);
  }

  // Enter / Leave bracket a caller's stay (SchedulerDecode takes Enter under the table lock, so that SchedulerShutdown never deletes a scheduler somebody is about to use)
  void Enter() { std::lock_guard<std::mutex> lock(mu_); callers_++; }
  void Leave() { std::lock_guard<std::mutex> lock(mu_); if (--callers_ == 0) idle_cv_.notify_all(); }

  int Decode(const uint8_t* data, size_t size, const OutputSpec& spec, void* dst, size_t dst_size, std::string* error) {
    Request r;
    r.data = data; r.size = size; r.spec = spec; r.spec.device_ptr = nullptr; r.dst = dst; r.dst_size = dst_size;
    const double t_in = Trace() ? NowMs() : 0;
    {
      std::unique_lock<std::mutex> lock(mu_);
      if (shutdown_) { if (error) *error = "scheduler shut down"; return 1; }
      pending_.push_back(&r);
      cv_.notify_all();
      r.cv.wait(lock, [&] { return r.done; });
    }
    int rc = r.status == 0 ? 0 : 1;
    const double t_woke = Trace() ? NowMs() : 0;
    if (rc == 0 && r.staging) memcpy(dst, r.staging, dst_size);     // every caller copies its own pixels: T threads, T copies at once
    if (Trace()) fprintf(stderr, "[req %.1f] waited %.1f ms, copy-out %.1f ms\n", NowMs(), t_woke - t_in, NowMs() - t_woke);
    if (r.staging) { std::lock_guard<std::mutex> lock(mu_); ReleaseStaging(r.staging, r.staging_cap); }
    if (rc && error) *error = r.error.empty() ? "decode failed" : r.error;
    return rc;
  }
  void Stats(int64_t* jobs, int64_t* images) { std::lock_guard<std::mutex> lock(mu_); if (jobs) *jobs = jobs_; if (images) *images = images_; }

 private:
  struct InFlight { int64_t ticket; std::vector<Request*> reqs; };

  // pinned staging buffers, recycled by capacity (mu_ held)
  void* TakeStaging(size_t bytes, size_t* cap) {
    int best = -1;
    for (size_t i = 0; i < free_staging_.size(); i++)
      if (free_staging_[i].second >= bytes && free_staging_[i].second <= 2 * bytes + (1 << 20) && (best < 0 || free_staging_[i].second < free_staging_[(size_t)best].second)) best = (int)i;
    if (best >= 0) { void* p = free_staging_[(size_t)best].first; *cap = free_staging_[(size_t)best].second; staging_held_ -= *cap; free_staging_.erase(free_staging_.begin() + best); return p; }
    return nullptr;
  }
  void ReleaseStaging(void* p, size_t cap) {
    static const size_t limit = (size_t)std::max(0, EnvInt("JXL_HIP_SCHED_PINNED_MB", 4096)) << 20;
    if (staging_held_ + cap <= limit && free_staging_.size() < 256) { free_staging_.push_back({p, cap}); staging_held_ += cap; }
    else { (void)hipSetDevice(device_); (void)hipHostFree(p); }
  }

  void Finish(Request* r, int status, const std::string& err) {     // (mu_ held)
    r->status = status; r->error = err; r->done = true;
    r->cv.notify_all();
  }

  void SubmitJob(std::vector<Request*>& reqs) {
    const double t_stage = NowMs();
    // staging for every request (pinned: the copy engine writes it while later jobs decode)
    std::vector<const uint8_t*> datas; std::vector<size_t> sizes, caps; std::vector<void*> outs;
    for (Request* r : reqs) {
      if (!r->staging) {
        size_t cap = 0;
        void* p;
        { std::lock_guard<std::mutex> lock(mu_); p = TakeStaging(r->dst_size, &cap); }
        if (!p) {
          cap = (r->dst_size + 4095) & ~(size_t)4095;
          if (hipHostMalloc(&p, cap, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); p = nullptr; }
        }
        r->staging = p; r->staging_cap = p ? cap : 0;
      }
      datas.push_back(r->data); sizes.push_back(r->size); outs.push_back(r->staging); caps.push_back(r->staging ? r->dst_size : 0);
    }
    int64_t ticket = -1;
    std::string err;
    const double t_sub = t_stage;
    try { ticket = pipe_->Submit(datas.data(), sizes.data(), (int)reqs.size(), reqs[0]->spec, nullptr, outs.data(), caps.data()); }
    catch (const std::exception& e) { err = e.what(); }
    std::lock_guard<std::mutex> lock(mu_);
    if (ticket < 0) { for (Request* r : reqs) Finish(r, 1, err); return; }
    if (Trace()) fprintf(stderr, "[sched %.1f] job %lld: %zu images submitted (staging + submit %.1f ms)\n", NowMs(), (long long)ticket, reqs.size(), NowMs() - t_sub);
    jobs_++; images_ += (int64_t)reqs.size();
    inflight_count_++;
    inflight_.push_back(InFlight{ticket, reqs});
    done_cv_.notify_all();
  }

  void CollectorLoop() {
    (void)hipSetDevice(device_);
    for (;;) {
      std::vector<Request*> reqs;
      {
        std::unique_lock<std::mutex> lock(mu_);
        // At most max_jobs_ jobs on their way: what arrives meanwhile waits and rides together.  A job's cost on the GPU hardly depends on its size (the entropy stages
        // are latency chains per stream), so a few large jobs in flight beat many small ones — measured with 64 callers: 0.9 Gpixel/s with a job per arrival.
        cv_.wait(lock, [&] { return shutdown_ || (!pending_.empty() && (int)inflight_count_ < max_jobs_); });
        if (shutdown_) { for (Request* r : pending_) Finish(r, 1, "scheduler shut down"); pending_.clear(); return; }
        // A window for company: threads that were released together by a job come back together, a few milliseconds apart (each copies its pixels out first), and one job
        // of 64 frames costs the GPU little more than one of a single frame (the entropy stages are latency chains per stream).  The job goes out when nothing new has
        // arrived for quiet_us, after max_wait_us at the latest, or when it is full.
        // (a lone caller — nobody else inside Decode, nothing in flight whose callers could come back — does not wait for company that cannot come)
        const bool alone = callers_ <= 1 && inflight_count_ == 0;
        // Largest job: a third of the callers inside Decode right now (at least 8, at most max_job_).  Callers are synchronous — each waits for its image —, so C callers are
        // C images in flight at most: as ONE job they pay a job's whole latency (LF, HF, pixels: ~140 ms at 64) per round; as three jobs a third apart the stages of one overlap
        // the next one's (64 callers: 3.0-3.3 Gpixel/s with jobs of up to 64, 3.8-4.3 with jobs of up to 22, four in flight: profiles/r06_notes.md section 15).
        // (the callers that exist = the decoders that exist: the reference crate's JxlDecoder owns one and is not shared between threads; the count of threads inside Decode
        // dips while they copy their pixels out and allocate the next buffer — a cap that followed it measured 2.7-3.8 against 4.1-4.3 Gpixel/s for the fixed 22)
        const int job_cap = max_job_fixed_ ? max_job_ : std::min(max_job_, std::max(8, (std::max(g_live_decoders.load(), callers_) + 2) / 3));
        if ((int)pending_.size() < job_cap && max_wait_us_ > 0 && !alone) {
          const auto t_first = std::chrono::steady_clock::now();
          auto t_last = t_first;
          size_t seen = pending_.size();
          while (!shutdown_ && (int)pending_.size() < job_cap) {
            const auto now = std::chrono::steady_clock::now();
            auto deadline = std::min(t_first + std::chrono::microseconds(max_wait_us_), t_last + std::chrono::microseconds(quiet_us_));
            // the callers of a job that has just completed are on their way back (each copies its pixels out first): the first of them does not leave alone
            if (pending_.size() * 4 < expected_back_ * 3 && now < expected_until_) deadline = std::max(deadline, std::min(expected_until_, t_first + std::chrono::microseconds(5 * max_wait_us_)));
            if (now >= deadline) break;
            cv_.wait_until(lock, deadline);
            if (pending_.size() != seen) { seen = pending_.size(); t_last = std::chrono::steady_clock::now(); }
          }
          expected_back_ = 0;
        }
        // the requests that share the first one's output format, in arrival order
        const OutputSpec spec = pending_.front()->spec;
        for (auto it = pending_.begin(); it != pending_.end() && (int)reqs.size() < job_cap;) {
          if (SameSpec((*it)->spec, spec)) { reqs.push_back(*it); it = pending_.erase(it); } else ++it;
        }
      }
      SubmitJob(reqs);
    }
  }

  void CompleterLoop() {
    (void)hipSetDevice(device_);
    for (;;) {
      InFlight job;
      {
        std::unique_lock<std::mutex> lock(mu_);
        done_cv_.wait(lock, [&] { return (shutdown_ && collector_done_) || !inflight_.empty(); });
        if (inflight_.empty()) { if (shutdown_ && collector_done_) return; continue; }
        job = inflight_.front(); inflight_.pop_front();
      }
      PipelineJobResult res;
      std::string err;
      try { pipe_->Wait(job.ticket, &res); } catch (const std::exception& e) { err = e.what(); }
      if (Trace()) fprintf(stderr, "[sched %.1f] job %lld done\n", NowMs(), (long long)job.ticket);
      std::vector<Request*> retry;
      {
        std::lock_guard<std::mutex> lock(mu_);
        inflight_count_--;
        if (job.reqs.size() >= 4) { expected_back_ = job.reqs.size(); expected_until_ = std::chrono::steady_clock::now() + std::chrono::microseconds(return_us_); }
        cv_.notify_all();
        for (size_t i = 0; i < job.reqs.size(); i++) {
          Request* r = job.reqs[i];
          if (!err.empty()) { Finish(r, 1, err); continue; }
          if (res.status[i] == 0) { Finish(r, 0, std::string()); continue; }
          // one image can fail a whole job in Prepare (a feature the device path refuses at that point): everybody gets a job of their own once
          const bool whole_job = job.reqs.size() > 1 && !r->retried && [&] { for (size_t k = 0; k < res.status.size(); k++) if (res.status[k] == 0) return false; return true; }();
          if (whole_job) { r->retried = true; retry.push_back(r); } else Finish(r, 1, res.error[i]);
        }
      }
      for (Request* r : retry) { std::vector<Request*> one{r}; SubmitJob(one); }
    }
  }

  const int device_;
  std::unique_ptr<Pipeline> pipe_;
  std::mutex mu_;
  std::condition_variable cv_, done_cv_;
  std::deque<Request*> pending_;
  std::deque<InFlight> inflight_;
  std::vector<std::pair<void*, size_t>> free_staging_;
  size_t staging_held_ = 0;
  int max_job_ = 64, quiet_us_ = 1000, max_wait_us_ = 8000, max_jobs_ = 4;
  bool max_job_fixed_ = false;            // JXL_HIP_SCHED_MAX_JOB given: no cap by the number of callers
  int inflight_count_ = 0;
  size_t expected_back_ = 0; std::chrono::steady_clock::time_point expected_until_{}; int return_us_ = 25000;
  int64_t jobs_ = 0, images_ = 0;
  bool shutdown_ = false, collector_done_ = false;
  int callers_ = 0;                       // threads between Enter and Leave
  std::condition_variable idle_cv_;
  std::thread collector_, completer_;
};

std::mutex g_sched_mu;
DeviceScheduler* g_sched[64] = {nullptr};       // (never destroyed at exit: the HIP runtime may be gone by then; SchedulerShutdown for tests)

}  // namespace

int SchedulerDecode(int device, const uint8_t* data, size_t size, const OutputSpec& spec, void* dst, size_t dst_size, std::string* error) {
  DeviceScheduler* s = nullptr;
  try {
    if (device >= 0 && device < 64) {
      std::lock_guard<std::mutex> lock(g_sched_mu);
      if (!g_sched[device]) g_sched[device] = new DeviceScheduler(device);
      s = g_sched[device];
      s->Enter();
    }
  } catch (const std::exception& e) { if (error) *error = e.what(); return 1; }
  if (!s) { if (error) *error = "no scheduler for this device"; return 1; }
  const int rc = s->Decode(data, size, spec, dst, dst_size, error);
  s->Leave();
  return rc;
}

void SchedulerStats(int device, int64_t* jobs, int64_t* images) {
  if (jobs) *jobs = 0;
  if (images) *images = 0;
  std::lock_guard<std::mutex> lock(g_sched_mu);
  if (device >= 0 && device < 64 && g_sched[device]) g_sched[device]->Stats(jobs, images);
}

void SchedulerNoteDecoder(int delta) { g_live_decoders.fetch_add(delta); }
void SchedulerShutdown() {     // legal at any time: requests that have not been submitted fail ("scheduler shut down"), the ones in flight complete, their callers leave, then the schedulers go
  std::lock_guard<std::mutex> lock(g_sched_mu);
  for (auto& s : g_sched) { delete s; s = nullptr; }
}

}  // namespace jxlhip
