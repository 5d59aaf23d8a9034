// jxl-hip: streaming decode pipeline of one GPU (pipeline.h).  Threads: `prepare_threads` workers (parse, tables, upload, LF stage), one issuer (HF stages and
// tails in submission order); Submit / Wait are called by anybody.  All GPU work goes to non-blocking streams of the object's own; nothing touches the NULL stream.
#include "pipeline.h"
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace jxlhip {

#define HIP_CHECK(expr)                                                                                    \
  do {                                                                                                     \
    hipError_t e_ = (expr);                                                                                \
    if (e_ != hipSuccess) throw ParseError(std::string("HIP error: ") + hipGetErrorString(e_) + " in " #expr, false); \
  } while (0)

namespace {
void* NewStream(int priority) {
  hipStream_t s = nullptr;
  HIP_CHECK(hipStreamCreateWithPriority(&s, hipStreamNonBlocking, priority));
  return s;
}
}  // namespace
// The k-th of the pipeline's (at most four) side streams for the inverse-transform chains of a job's Modular images (Batch::EnqueueModularTail), made when first asked for:
// pipelines that never see such jobs — the one behind the libjxl API decodes single images — keep the process's stream count where it was.
void* Pipeline::TailStream(int k) {
  static const int cap = getenv("JXL_HIP_MOD_TAIL_STREAMS") ? std::max(1, std::min(16, atoi(getenv("JXL_HIP_MOD_TAIL_STREAMS")))) : 4;
  if (k < 0 || k >= cap || cap < 2) return nullptr;
  std::lock_guard<std::mutex> lock(tail_mu_);
  while ((int)tail_side_.size() <= k) tail_side_.push_back(NewStream(0));
  return tail_side_[(size_t)k];
}
namespace {
void* NewEvent(bool timing = false) {
  hipEvent_t e = nullptr;
  HIP_CHECK(hipEventCreateWithFlags(&e, timing ? hipEventDefault : hipEventDisableTiming));
  return e;
}
void Record(void* ev, void* stream) { HIP_CHECK(hipEventRecord((hipEvent_t)ev, (hipStream_t)stream)); }
void StreamWait(void* stream, void* ev) { HIP_CHECK(hipStreamWaitEvent((hipStream_t)stream, (hipEvent_t)ev, 0)); }
size_t Align256(size_t v) { return (v + 255) / 256 * 256; }
}  // namespace

// The caller's current device comes back at the end of every public entry point (a thread that drives pipelines on several GPUs, or mixes its own HIP code
// with the pipeline, must not find its device changed after a Submit or a Wait).  The pipeline's own threads stay on their device.
namespace {
struct DeviceScope {
  int prev = -1;
  explicit DeviceScope(int device) { if (hipGetDevice(&prev) != hipSuccess) prev = -1; if (prev != device) (void)hipSetDevice(device); else prev = -1; }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};
}  // namespace

Pipeline::Pipeline(int device, const PipelineOptions& opt) : device_(device), opt_(opt) {
  DeviceScope scope(device_);
  opt_.in_flight = std::max(1, std::min(opt_.in_flight, 64));
  opt_.hf_streams = std::max(1, std::min(opt_.hf_streams, 12));
  opt_.lf_streams = std::max(1, std::min(opt_.lf_streams, 32));
  opt_.prepare_threads = std::max(1, std::min(opt_.prepare_threads, 16));
  opt_.parse_threads = std::max(1, std::min(opt_.parse_threads, 64));
  ncoef_ = opt_.hf_streams + 1;                  // one coefficient set per HF stage in flight + the one the tail is consuming
  // batch objects: the jobs in flight on the GPU + the ones the prepare threads are filling (+ one: a worker waits for its slot's previous job to leave the GPU)
  nbuf_ = opt_.in_flight + 2 + 1;
  if (nbuf_ % ncoef_) nbuf_ += ncoef_ - nbuf_ % ncoef_;     // (job k always meets coefficient set k % ncoef and slot k % nbuf)
  // the few long wavefronts of the entropy stages on high-priority streams, everything else at the default priority (0; the range's other end would be "low")
  int prio_least = 0, prio_high = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_high);   // (numerically lower = higher priority)
  prio_high = std::min(prio_high, 0);
  main_ = NewStream(0);
  d2h_[0] = NewStream(0); d2h_[1] = NewStream(0);
  for (int i = 0; i < opt_.lf_streams; i++) lf_side_.push_back(NewStream(prio_high));
  for (int i = 0; i < opt_.hf_streams; i++) hf_side_.push_back(NewStream(prio_high));
  clock_event_ = NewEvent(true);
  Record(clock_event_, main_);
  coef_.assign((size_t)ncoef_, SharedPlanes());
  for (int b = 0; b < nbuf_; b++) {
    std::unique_ptr<Slot> s(new Slot());
    s->batch.reset(new Batch(device_));
    s->batch->SetTailStreams([this](int k) -> void* { return TailStream(k); });
    s->lf_done = NewEvent(); s->front_done = NewEvent(); s->hf_done = NewEvent(); s->idct_done = NewEvent(); s->rest_done = NewEvent();
    slots_.push_back(std::move(s));
  }
  if (opt_.reserve_frames > 0 && opt_.reserve_width > 0 && opt_.reserve_height > 0) {
    // layout of Batch::Prepare for plain VarDCT frames: per frame and channel num_groups x 65536 int32 coefficients, one padded float plane
    const size_t ng = (size_t)((opt_.reserve_width + 255) / 256) * ((opt_.reserve_height + 255) / 256);
    const size_t bw = (size_t)(opt_.reserve_width + 7) / 8, bh = (size_t)(opt_.reserve_height + 7) / 8;
    want_coef_ = (size_t)opt_.reserve_frames * 3 * ng * 65536 * 4;
    want_big_ = (size_t)opt_.reserve_frames * 3 * Align256(bw * 8 * bh * 8 * 4) * (size_t)std::max(1, std::min(opt_.reserve_plane_sets, 2));
    ReservePlanes(&big_, want_big_);
    for (auto& c : coef_) ReservePlanes(&c, want_coef_);
  }
  for (int w = 0; w < opt_.prepare_threads; w++) workers_.emplace_back([this, w] { PrepareWorker(w); });
  issuer_ = std::thread([this] { IssuerLoop(); });
}

Pipeline::~Pipeline() {
  { std::lock_guard<std::mutex> lock(mu_); shutdown_ = true; }
  cv_.notify_all();
  for (auto& t : workers_) t.join();
  if (issuer_.joinable()) issuer_.join();
  DeviceScope scope(device_);
  for (void* s : lf_side_) (void)hipStreamSynchronize((hipStream_t)s);
  for (void* s : hf_side_) (void)hipStreamSynchronize((hipStream_t)s);
  (void)hipStreamSynchronize((hipStream_t)main_);
  for (void* s : d2h_) (void)hipStreamSynchronize((hipStream_t)s);
  for (auto& kv : jobs_) if (kv.second->done_event) (void)hipEventDestroy((hipEvent_t)kv.second->done_event);
  jobs_.clear();
  for (auto& s : slots_) {
    s->batch.reset();
    for (void* e : {s->lf_done, s->front_done, s->hf_done, s->idct_done, s->rest_done}) if (e) (void)hipEventDestroy((hipEvent_t)e);
  }
  slots_.clear();
  if (big_.p) DeviceArenaGive(big_.p, big_.cap, device_);
  for (auto& c : coef_) if (c.p) DeviceArenaGive(c.p, c.cap, device_);
  if (clock_event_) (void)hipEventDestroy((hipEvent_t)clock_event_);
  for (void* s : lf_side_) (void)hipStreamDestroy((hipStream_t)s);
  for (void* s : hf_side_) (void)hipStreamDestroy((hipStream_t)s);
  for (void* s : tail_side_) { (void)hipStreamSynchronize((hipStream_t)s); (void)hipStreamDestroy((hipStream_t)s); }
  (void)hipStreamDestroy((hipStream_t)main_);
  for (void* s : d2h_) (void)hipStreamDestroy((hipStream_t)s);
}

// (device idle as far as these planes go) makes sp a block of at least `bytes`: from the arena pool, else from the runtime
void Pipeline::ReservePlanes(SharedPlanes* sp, size_t bytes) {
  if (sp->p && sp->cap >= bytes) return;
  if (sp->p) { DeviceArenaGive(sp->p, sp->cap, device_, /*idle=*/true); sp->p = nullptr; sp->cap = 0; }
  if (!bytes) return;
  const size_t want = bytes + bytes / 16;
  size_t cap = 0;
  void* p = DeviceArenaTake(want, &cap, device_);
  if (!p) {
    if (hipMalloc(&p, want) != hipSuccess) {
      (void)hipGetLastError();
      DeviceArenaPoolTrim();
      HIP_CHECK(hipMalloc(&p, want));
    }
    cap = want;
  }
  sp->p = (uint8_t*)p; sp->cap = cap; sp->dirty = true; sp->clean_extent = 0;
}

void Pipeline::GrowSharedWhenIdle() {
  // (mu_ held; every job submitted so far has been harvested: nothing on the GPU refers to the shared planes)
  if (want_big_ > big_.cap) ReservePlanes(&big_, want_big_);
  for (auto& c : coef_) if (want_coef_ > c.cap) ReservePlanes(&c, want_coef_);
}

Pipeline::Job* Pipeline::FindJob(int64_t ticket) {
  auto it = jobs_.find(ticket);
  return it == jobs_.end() ? nullptr : it->second.get();
}

int64_t Pipeline::Submit(const uint8_t* const* datas, const size_t* sizes, int n, const OutputSpec& spec, void* const* device_out, void* const* host_out, const size_t* out_capacity) {
  if (n <= 0 || !datas || !sizes) throw ParseError("JxlHipPipelineSubmit: no images", false);
  if (!device_out == !host_out) throw ParseError("JxlHipPipelineSubmit: exactly one of device_out / host_out must be given", false);
  std::shared_ptr<Job> job(new Job());
  job->datas.assign(datas, datas + n); job->sizes.assign(sizes, sizes + n);
  if (device_out) job->device_out.assign(device_out, device_out + n);
  if (host_out) job->host_out.assign(host_out, host_out + n);
  if (out_capacity) job->capacity.assign(out_capacity, out_capacity + n);
  job->spec = spec; job->spec.device_ptr = nullptr;
  job->result.status.assign((size_t)n, 1); job->result.error.assign((size_t)n, std::string());
  std::unique_lock<std::mutex> lock(mu_);
  if (shutdown_) throw ParseError("JxlHipPipelineSubmit: the pipeline is shutting down", false);
  // back-pressure: at most nbuf - 1 jobs between submission and the issue of their tail (the issuer's progress never depends on anybody calling Wait)
  cv_.wait(lock, [&] { return shutdown_ || next_ticket_ - next_issue_ < nbuf_ - 1; });
  if (shutdown_) throw ParseError("JxlHipPipelineSubmit: the pipeline is shutting down", false);
  bool idle = next_issue_ == next_ticket_;
  if (idle) for (auto& kv : jobs_) if (kv.second->state != kHarvested) { idle = false; break; }
  if (idle) {
    DeviceScope scope(device_);
    GrowSharedWhenIdle();
    cold_count_ = 0;
  }
  job->ticket = next_ticket_++;
  {
    // the first jobs of a cold pipeline take the one-wavefront-per-stream LF kernel — one after the other: job k's LF stage waits for job k - 1's (on the device), so that the first
    // job's front is done after one kernel time instead of all of them after four (small jobs — latency mode — are wide too, but never chained: they overlap)
    const bool cold = cold_count_ < opt_.wide_first && n > opt_.small_job_frames;
    static const bool no_chain = getenv("JXL_HIP_NO_WIDE_CHAIN") != nullptr;
    job->wide_chain = cold && !no_chain;
    job->wide_after = job->wide_chain && cold_count_ > 0 ? job->ticket - 1 : -1;
    job->cold_wide = cold_count_++ < opt_.wide_first || n <= opt_.small_job_frames;
  }
  // results nobody collects: keep a bounded history
  while (jobs_.size() > (size_t)(8 * nbuf_ + 64)) {
    auto it = jobs_.begin();
    if (it->second->state != kHarvested) break;
    if (it->second->done_event) (void)hipEventDestroy((hipEvent_t)it->second->done_event);
    jobs_.erase(it);
  }
  jobs_[job->ticket] = job;
  prep_queue_.push_back(job);
  lock.unlock();
  cv_.notify_all();
  return job->ticket;
}

void Pipeline::PrepareWorker(int worker) {
  (void)hipSetDevice(device_);
  for (;;) {
    std::shared_ptr<Job> job;
    {
      std::unique_lock<std::mutex> lock(mu_);
      cv_.wait(lock, [&] { return shutdown_ || !prep_queue_.empty(); });
      if (shutdown_) return;
      job = prep_queue_.front(); prep_queue_.pop_front();
      job->state = kPreparing;
    }
    // the slot's previous job must have left the GPU: its batch object is about to be refilled
    if (job->ticket >= nbuf_) {
      std::shared_ptr<Job> prev;
      {
        std::unique_lock<std::mutex> lock(mu_);
        const int64_t pt = job->ticket - nbuf_;
        cv_.wait(lock, [&] { if (shutdown_) return true; auto it = jobs_.find(pt); return it == jobs_.end() || it->second->state >= kTailIssued; });
        if (shutdown_) return;
        auto it = jobs_.find(pt);
        if (it != jobs_.end()) prev = it->second;
      }
      if (prev) Harvest(prev.get());
      else {
        // (collected and dropped already: its Wait harvested it)
      }
    }
    PrepareJob(job.get(), worker);
    { std::lock_guard<std::mutex> lock(mu_); job->state = kFrontIssued; if (job->wide_chain) wide_enqueued_ = std::max(wide_enqueued_, job->ticket); /* (also when its prepare failed: the job behind it must not wait for ever) */ }
    cv_.notify_all();
  }
}

void Pipeline::PrepareJob(Job* j, int worker) {
  const auto t0 = std::chrono::steady_clock::now();
  auto t_parse = t0, t_prep = t0;
  Slot& s = *slots_[(size_t)(j->ticket % nbuf_)];
  Batch& bt = *s.batch;
  const int n = (int)j->datas.size();
  try {
    HIP_CHECK(hipSetDevice(device_));
    bt.Reset();
    vec<int> index;
    std::vector<std::string> errors;
    bt.AddImagesTolerant(j->datas.data(), j->sizes.data(), n, opt_.parse_threads, &index, &errors);
    t_parse = std::chrono::steady_clock::now();
    j->batch_index.assign(index.begin(), index.end());
    bool any = false;
    for (int i = 0; i < n; i++) {
      if (index[(size_t)i] < 0) { j->result.error[(size_t)i] = errors[(size_t)i]; continue; }
      OutputSpec o = j->spec;
      o.device_ptr = j->device_out.empty() ? nullptr : j->device_out[(size_t)i];
      const size_t need = bt.OutputSizeOf(index[(size_t)i], o);
      if (!j->capacity.empty() && j->capacity[(size_t)i] < need) {
        // (the image stays in the batch — taking it out would renumber the others — and decodes into a buffer of the batch's own; it is reported as failed)
        j->result.error[(size_t)i] = "output buffer too small for this image";
        o.device_ptr = nullptr;
        j->batch_index[(size_t)i] = -2 - index[(size_t)i];
      } else if ((!j->device_out.empty() && !j->device_out[(size_t)i]) || (!j->host_out.empty() && !j->host_out[(size_t)i])) {
        j->result.error[(size_t)i] = "no output buffer for this image";
        o.device_ptr = nullptr;
        j->batch_index[(size_t)i] = -2 - index[(size_t)i];
      } else any = true;
      bt.SetOutput(index[(size_t)i], o);
    }
    if (!any) { j->job_error = "no image of the job could be parsed"; return; }
    bt.cfg.lane_stride_lf = opt_.lane_stride_lf; bt.cfg.lane_stride_hf = opt_.lane_stride_hf; bt.cfg.no_flag_wait = opt_.no_flag_wait;
    // latency mode: sparse wavefronts in the SIMT HF stage — one group stream per wavefront for a handful of frames, four up to a hundred (kernels.hip LaunchHfDecode)
    bt.cfg.hf_lanes_per_wave = bt.cfg.hf_lanes_per_wg = 0;
    if (opt_.hf_sparse || n <= opt_.small_job_frames) {
      if (n <= 8) { bt.cfg.hf_lanes_per_wave = 1; bt.cfg.hf_lanes_per_wg = 16; }
      else if (n <= 96) { bt.cfg.hf_lanes_per_wave = 4; bt.cfg.hf_lanes_per_wg = 64; }
    }
    bt.UseSharedPlanes(&big_, &coef_[(size_t)(j->ticket % ncoef_)]);
    // tables + upload go to the stream the job's LF stage runs on, which follows without a host-side wait (an upload stream of its own was seen waiting tens of
    // milliseconds behind other streams' entropy kernels)
    void* side = lf_side_[(size_t)(j->ticket % (int64_t)lf_side_.size())];
    bt.Prepare(side, /*wait_upload=*/false);
    t_prep = std::chrono::steady_clock::now();
    {
      std::lock_guard<std::mutex> lock(mu_);
      want_big_ = std::max(want_big_, bt.big_bytes_wanted()); want_coef_ = std::max(want_coef_, bt.coef_bytes_wanted());
      if (!bt.uses_shared_big() || !bt.uses_shared_coef()) private_plane_jobs_++;
      bt.StageBytes(last_stage_bytes_);
      for (const char* k : {"lf_simt_frames", "lf_legacy_frames", "lf_simt_wp", "lf_simt_lanes", "lf_simt_waves"}) last_info_[k] = bt.Info(k);
      last_info_["compressed_bytes"] = (int64_t)bt.compressed_bytes();
      last_info_["total_pixels"] = (int64_t)bt.total_pixels();
      last_info_["frames"] = (int64_t)bt.size();
    }
    // the LF stage goes out right away, from this thread: the earlier it starts the better
    if (j->cold_wide && opt_.lane_stride_lf < 64) bt.cfg.lf_wide_once = 1;
    // (jobs whose LF trees use the weighted predictor keep the SIMT kernel whatever lf_wide_once says — LaunchLfDecode —: 420 ms a launch, nothing to chain)
    static const bool wide_wp = getenv("JXL_HIP_WIDE_WP") != nullptr;       // experiments (kernels.hip LaunchLfDecode)
    const bool lf_wide = bt.cfg.lf_wide_once && (wide_wp || !bt.Info("lf_simt_wp"));
    if (j->wide_chain) { std::lock_guard<std::mutex> lock(mu_); j->lf_wide = lf_wide; }
    if (j->wide_after >= 0 && lf_wide) {
      // behind the cold-start job before it: wait (host) until that job's LF stage is in its stream, then make ours wait for it (device)
      std::unique_lock<std::mutex> lock(mu_);
      cv_.wait(lock, [&] { return shutdown_ || wide_enqueued_ >= j->wide_after; });
      auto it = jobs_.find(j->wide_after);
      const bool chained = it != jobs_.end() && it->second->job_error.empty() && it->second->state < kHarvested && it->second->lf_wide;
      lock.unlock();
      if (chained) StreamWait(side, slots_[(size_t)(j->wide_after % nbuf_)]->lf_done);
    }
    bt.RunPart(side, 5, opt_.timed != 0);           // LF decode + varblock placement: all the HF stage waits for
    Record(s.lf_done, side);
    if (j->wide_chain) { { std::lock_guard<std::mutex> lock(mu_); wide_enqueued_ = std::max(wide_enqueued_, j->ticket); } cv_.notify_all(); }
    bt.RunPart(side, 6, opt_.timed != 0);           // LF post-processing: needed by the IDCT only
    Record(s.front_done, side);
  } catch (const std::exception& e) {
    j->job_error = e.what();
    if (j->job_error.empty()) j->job_error = "prepare failed";
  }
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  static const bool trace = getenv("JXL_HIP_SCHED_TRACE") != nullptr;
  if (trace) fprintf(stderr, "[pipe %.1f] job %lld (%d images, slot %d): prepare + LF enqueue %.1f ms (parse %.1f, prepare %.1f)%s%s\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(),
                     (long long)j->ticket, n, (int)(j->ticket % nbuf_), dt * 1e3, std::chrono::duration<double, std::milli>(t_parse - t0).count(), std::chrono::duration<double, std::milli>(t_prep - t_parse).count(), j->job_error.empty() ? "" : " ERROR ", j->job_error.c_str());
  std::lock_guard<std::mutex> lock(mu_);
  prepare_s_total_ += dt; prepared_jobs_++;
}

void Pipeline::IssueHf(Job* j) {
  if (j->hf_issued) return;
  j->hf_issued = true;
  if (!j->job_error.empty()) return;
  Slot& s = *slots_[(size_t)(j->ticket % nbuf_)];
  void* hs = hf_side_[(size_t)(j->ticket % (int64_t)hf_side_.size())];
  try {
    StreamWait(hs, s.lf_done);
    // the coefficient set's previous user has consumed (and zeroed) it: its tail was issued before this call (tails go out in order, HF stages at most hf_streams ahead)
    if (j->ticket >= ncoef_) StreamWait(hs, slots_[(size_t)((j->ticket - ncoef_) % nbuf_)]->idct_done);
    s.batch->RunPart(hs, 3, opt_.timed != 0);
    Record(s.hf_done, hs);
  } catch (const std::exception& e) { j->job_error = e.what(); }
}

void Pipeline::IssueTail(Job* j) {
  Slot& s = *slots_[(size_t)(j->ticket % nbuf_)];
  Batch& bt = *s.batch;
  if (j->job_error.empty()) {
    try {
      StreamWait(main_, s.hf_done);
      StreamWait(main_, s.front_done);
      bt.RunPart(main_, 7, opt_.timed != 0);      // IDCT: the last stage that touches the coefficient set
      Record(s.idct_done, main_);
      bt.RunPart(main_, 8, opt_.timed != 0);      // restoration filters, colour, write
      Record(s.rest_done, main_);
      void* d2h = d2h_[0];
      StreamWait(d2h, s.rest_done);
      if (!j->host_out.empty()) {
        // equally sized images at equal distances on both sides (a job of same-shaped frames into one host buffer): one pitched copy instead of one per image
        bool pitched = j->host_out.size() >= 2;
        ptrdiff_t hp = 0, dp = 0;
        size_t osz = 0;
        for (size_t i = 0; i < j->host_out.size() && pitched; i++) {
          const int bi = j->batch_index[i];
          if (bi < 0) { pitched = false; break; }
          if (i == 0) { osz = bt.image(bi).out_size; continue; }
          const ptrdiff_t h = (const uint8_t*)j->host_out[i] - (const uint8_t*)j->host_out[i - 1];
          const ptrdiff_t d = (const uint8_t*)bt.device_output(bi) - (const uint8_t*)bt.device_output(j->batch_index[i - 1]);
          if (i == 1) { hp = h; dp = d; }
          if (h != hp || d != dp || bt.image(bi).out_size != osz || hp < (ptrdiff_t)osz || dp < (ptrdiff_t)osz) pitched = false;
        }
        static const bool no_pitched = getenv("JXL_HIP_NO_PITCHED_D2H") != nullptr;
        if (pitched && !no_pitched) {
          HIP_CHECK(hipMemcpy2DAsync(j->host_out[0], (size_t)hp, bt.device_output(j->batch_index[0]), (size_t)dp, osz, j->host_out.size(), hipMemcpyDeviceToHost, (hipStream_t)d2h));
        } else {
          for (size_t i = 0; i < j->host_out.size(); i++) {
            const int bi = j->batch_index[i];
            if (bi < 0) continue;
            HIP_CHECK(hipMemcpyAsync(j->host_out[i], bt.device_output(bi), bt.image(bi).out_size, hipMemcpyDeviceToHost, (hipStream_t)d2h));
          }
        }
      }
      bt.EnqueueStatusReadback(d2h);
    } catch (const std::exception& e) {
      j->job_error = e.what();
      // whatever was enqueued keeps the slot busy: Harvest drains the streams before the batch object is touched again
    }
  } else {
    // (nothing was enqueued beyond the front — if that; the events of this slot keep the state of its previous job, which completed long ago)
    try { Record(s.idct_done, main_); Record(s.rest_done, main_); } catch (...) {}
  }
  try {
    if (!j->done_event) j->done_event = NewEvent(true);
    Record(j->done_event, d2h_[0]);
  } catch (const std::exception& e) { if (j->job_error.empty()) j->job_error = e.what(); }
}

void Pipeline::IssuerLoop() {
  (void)hipSetDevice(device_);
  for (;;) {
    std::shared_ptr<Job> job;
    std::vector<std::shared_ptr<Job>> ahead;
    {
      std::unique_lock<std::mutex> lock(mu_);
      cv_.wait(lock, [&] { if (shutdown_) return true; auto it = jobs_.find(next_issue_); return it != jobs_.end() && it->second->state >= kFrontIssued; });
      if (shutdown_) return;
      job = jobs_[next_issue_];
      for (int a = 1; a <= opt_.hf_streams; a++) {
        auto it = jobs_.find(next_issue_ + a);
        if (it == jobs_.end() || it->second->state < kFrontIssued) break;       // (its LF stage is not enqueued yet: the events it would wait for are a previous job's)
        ahead.push_back(it->second);
      }
    }
    IssueHf(job.get());
    for (auto& a : ahead) IssueHf(a.get());
    IssueTail(job.get());
    { std::lock_guard<std::mutex> lock(mu_); job->state = kTailIssued; next_issue_++; }
    cv_.notify_all();
  }
}

void Pipeline::Harvest(Job* j) {
  Slot& s = *slots_[(size_t)(j->ticket % nbuf_)];
  std::lock_guard<std::mutex> slot_lock(s.mu);
  { std::lock_guard<std::mutex> lock(mu_); if (j->state >= kHarvested) return; }
  DeviceScope scope(device_);
  Batch& bt = *s.batch;
  const size_t n = j->datas.size();
  bool readback_ok = false;
  vec<uint32_t> st;
  if (j->job_error.empty()) {
    try { bt.HarvestStatus(&st); readback_ok = true; } catch (const std::exception& e) { j->job_error = e.what(); }
  }
  if (j->done_event) {
    (void)hipEventSynchronize((hipEvent_t)j->done_event);
    float ms = 0;
    if (hipEventElapsedTime(&ms, (hipEvent_t)clock_event_, (hipEvent_t)j->done_event) == hipSuccess) j->result.end_ms = ms; else (void)hipGetLastError();
  }
  if (!j->job_error.empty() && !readback_ok) {
    // a job that failed half-way: whatever it enqueued must have left the GPU before the slot is reused
    (void)hipStreamSynchronize((hipStream_t)main_);
    for (void* x : lf_side_) (void)hipStreamSynchronize((hipStream_t)x);
    for (void* x : hf_side_) (void)hipStreamSynchronize((hipStream_t)x);
    for (void* x : d2h_) (void)hipStreamSynchronize((hipStream_t)x);
  }
  for (size_t i = 0; i < n; i++) {
    const int bi = i < j->batch_index.size() ? j->batch_index[i] : -1;
    if (!j->result.error[i].empty()) { j->result.status[i] = 1; continue; }
    if (!j->job_error.empty() || bi < 0) { j->result.status[i] = 1; j->result.error[i] = j->job_error.empty() ? "not decoded" : j->job_error; continue; }
    uint32_t bad = 0;
    for (int u = bt.first_unit_of(bi); u < bt.first_unit_of(bi) + bt.num_units_of(bi); u++) bad |= st[(size_t)u];
    if (bad) {
      j->result.status[i] = 1;
      j->result.error[i] = (bad & kErrUnsupported) ? "unsupported: stream feature on the device path" : "corrupt stream (device status " + std::to_string(bad) + ")";
    } else j->result.status[i] = 0;
  }
  static const bool trace = getenv("JXL_HIP_SCHED_TRACE") != nullptr;
  if (trace && opt_.timed && readback_ok) {
    float tl[9];
    bt.DebugTimeline(clock_event_, tl);
    fprintf(stderr, "[pipe] job %lld timeline (ms): LF %.1f .. %.1f, LF post end %.1f, HF %.1f .. %.1f, IDCT %.1f .. %.1f, filters end %.1f, out end %.1f\n", (long long)j->ticket, tl[0], tl[1], tl[2], tl[7], tl[3],
            tl[8], tl[4], tl[5], tl[6]);
    int runs = 0;
    const StageTimes t = bt.CollectTimes(&runs);
    fprintf(stderr, "[pipe] job %lld (%zu images) on the GPU: lf %.1f lfpost %.1f hf %.1f idct %.1f filters %.1f out %.1f, first kernel to last %.1f ms; done at %.1f ms\n", (long long)j->ticket, n, t.lf_ms, t.lfpost_ms,
            t.hf_ms, t.idct_ms, t.filter_ms, t.out_ms, t.total_ms, j->result.end_ms);
  }
  {
    std::lock_guard<std::mutex> lock(mu_);
    if (readback_ok) last_info_["hf_nonzeros"] = bt.Info("hf_nonzeros");
    j->state = kHarvested;
  }
  cv_.notify_all();
}

void Pipeline::Wait(int64_t ticket, PipelineJobResult* out) {
  std::shared_ptr<Job> job;
  {
    std::unique_lock<std::mutex> lock(mu_);
    auto it = jobs_.find(ticket);
    if (it == jobs_.end()) throw ParseError("JxlHipPipelineWait: unknown ticket (never submitted, collected already, or too old)", false);
    job = it->second;
    if (job->waited) throw ParseError("JxlHipPipelineWait: this ticket is being waited for already", false);
    job->waited = true;
    cv_.wait(lock, [&] { return shutdown_ || job->state >= kTailIssued; });
    if (job->state < kTailIssued) throw ParseError("JxlHipPipelineWait: the pipeline is shutting down", false);
  }
  Harvest(job.get());
  if (out) *out = job->result;
  std::lock_guard<std::mutex> lock(mu_);
  if (job->done_event) { (void)hipEventDestroy((hipEvent_t)job->done_event); job->done_event = nullptr; }
  jobs_.erase(ticket);
}

void Pipeline::WaitAll() {
  for (;;) {
    std::shared_ptr<Job> job;
    {
      std::unique_lock<std::mutex> lock(mu_);
      const int64_t upto = next_ticket_;
      cv_.wait(lock, [&] { return shutdown_ || next_issue_ >= upto; });
      for (auto& kv : jobs_) if (kv.second->state == kTailIssued) { job = kv.second; break; }
      if (!job) return;
    }
    Harvest(job.get());
  }
}

void Pipeline::ResetClock() {
  WaitAll();
  std::lock_guard<std::mutex> lock(mu_);
  DeviceScope scope(device_);
  (void)hipStreamSynchronize((hipStream_t)main_);
  Record(clock_event_, main_);
  (void)hipStreamSynchronize((hipStream_t)main_);
  prepare_s_total_ = 0; prepared_jobs_ = 0;
}

StageTimes Pipeline::CollectTimes(int* runs) {
  WaitAll();
  DeviceScope scope(device_);
  StageTimes t; int total = 0;
  for (auto& s : slots_) {
    int r = 0;
    const StageTimes x = s->batch->CollectTimes(&r);
    total += r;
    t.lf_ms += x.lf_ms; t.lfpost_ms += x.lfpost_ms; t.hf_ms += x.hf_ms; t.idct_ms += x.idct_ms; t.filter_ms += x.filter_ms; t.out_ms += x.out_ms; t.total_ms += x.total_ms;
  }
  if (runs) *runs = total;
  return t;
}

void Pipeline::StageBytes(uint64_t out[6]) { std::lock_guard<std::mutex> lock(mu_); for (int i = 0; i < 6; i++) out[i] = last_stage_bytes_[i]; }

int64_t Pipeline::Info(const char* name) {
  const std::string n(name ? name : "");
  std::lock_guard<std::mutex> lock(mu_);
  if (n == "jobs") return next_ticket_;
  if (n == "slots") return nbuf_;
  if (n == "coefficient_sets") return ncoef_;
  if (n == "shared_big_bytes") return (int64_t)big_.cap;
  if (n == "shared_coef_bytes") { int64_t v = 0; for (auto& c : coef_) v += (int64_t)c.cap; return v; }
  if (n == "private_plane_jobs") return private_plane_jobs_;
  if (n == "device_bytes") {
    int64_t v = (int64_t)big_.cap;
    for (auto& c : coef_) v += (int64_t)c.cap;
    for (auto& s : slots_) v += (int64_t)(s->batch->const_bytes() + s->batch->work_bytes());
    return v;
  }
  auto it = last_info_.find(n);
  return it == last_info_.end() ? -1 : it->second;
}

}  // namespace jxlhip
