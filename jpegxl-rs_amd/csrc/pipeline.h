// jxl-hip: the streaming decode pipeline of one GPU as a library object (C ABI: JxlHipPipeline*, include/jxl_hip.h).
//
// A decode is LF (entropy decode of the LF groups: a serial chain per stream, hundreds of milliseconds per launch whatever the batch size) ->
// varblock placement + LF post-processing -> HF (entropy decode of the coefficients) -> IDCT -> filters + write.  Throughput comes from jobs in
// flight: while the tail (IDCT, filters, write) of job k runs on the main stream, the HF stage of job k + 1 runs on a stream of its own against
// the other coefficient set, the LF stages of the jobs behind it on side streams, and host worker threads parse, build tables for, upload and
// enqueue the LF stage of the jobs after those.  The object owns what that takes: the ring of batch objects (their LF-stage outputs), the
// coefficient sets and the pixel planes all jobs share, the HIP streams and events, the prepare threads and the thread that issues HF stages
// and tails in submission order.  (Rounds 1-4 kept this schedule in bench.py; it is a property of the library now.)
//
// The reference decodes batches by looping decode_with over files (jpegxl-rs/benches/decode.rs:16-19): there is no counterpart of this object
// in the reference API.  The libjxl-compatible JxlDecoder reaches it through DeviceScheduler (scheduler.cc): concurrent decode_with callers of
// one process are coalesced into jobs of one shared pipeline per device.
#pragma once
#include "decoder.h"
#include <atomic>
#include <condition_variable>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace jxlhip {

struct PipelineOptions {
  int in_flight = 11;        // jobs in flight on the GPU (LF stages run this many jobs ahead of the tail, minus one)
  int lf_streams = 11;       // side streams the LF stages are spread over (one per job in flight: weighted-predictor LF stages take ~590 ms per launch against steps of ~75 ms, seven streams bound such frames at 84 ms per step)
  int hf_streams = 2;        // HF stages in flight beside the tail (one stream and one coefficient set each, + the set the tail consumes)
  int prepare_threads = 3;   // host threads that each parse + prepare + upload one job at a time and enqueue its LF stage
  int parse_threads = 8;     // host threads one job's images are parsed on
  int lane_stride_lf = 8, lane_stride_hf = 1;
  int wide_first = 4;        // LF stages at the start of a cold pipeline that take the one-wavefront-per-stream kernel (shorter latency on an idle GPU)
  int hf_sparse = 0;         // latency mode: small jobs spread their HF group streams over many sparse wavefronts (1 per wavefront up to 8 frames, 4 up to 96)
  int small_job_frames = 0;  // jobs of at most this many frames always take it (latency mode: DeviceScheduler)
  int no_flag_wait = 0;      // the issuing thread never waits for a job's LF stage (placement flags): every IDCT kernel variant is launched instead — latency mode
  int timed = 0;             // bracket the stages with HIP events (CollectTimes)
  // shared coefficient sets / pixel planes are sized for jobs of this shape at creation (0: nothing is reserved; jobs run on arenas of their own until the
  // pipeline is idle, then the shared planes grow to the largest job seen)
  int reserve_frames = 0, reserve_width = 0, reserve_height = 0;
  int reserve_plane_sets = 1;   // 2: frames that take the stage-by-stage filters (anything but gaborish + one EPF pass) need a second set of pixel planes
};

struct PipelineJobResult {
  std::vector<int> status;            // per image: 0 decoded, 1 failed
  std::vector<std::string> error;     // per image: "" or what went wrong
  float end_ms = 0;                   // when the job's last byte had been written / copied, ms after the pipeline's clock was reset
};

class Pipeline {
 public:
  Pipeline(int device, const PipelineOptions& opt);
  ~Pipeline();
  // Submits a job of n compressed images, all decoded to `spec`.  Exactly one of device_out / host_out is given: n caller-owned destinations, each at least as large
  // as the image's output in that format (out_capacity[i], when given, is checked).  Host destinations should be pinned (hipHostMalloc / JxlHipHostAlloc) — pageable
  // memory works, slower.  The compressed bytes and the destinations must stay valid until Wait(ticket) returns.  Blocks while `in_flight` jobs are on their way.
  // Returns the job's ticket (>= 0).
  int64_t Submit(const uint8_t* const* datas, const size_t* sizes, int n, const OutputSpec& spec, void* const* device_out, void* const* host_out, const size_t* out_capacity);
  // Waits until job `ticket` has left the GPU (outputs written, host copies done) and hands out its per-image results; a ticket can be waited for once.
  void Wait(int64_t ticket, PipelineJobResult* out);
  void WaitAll();                                  // every job submitted so far has left the GPU (results stay collectable)
  void ResetClock();                               // end_ms of later jobs counts from here (call on an idle pipeline)
  StageTimes CollectTimes(int* runs);              // per-stage HIP-event sums over all jobs since the last call (options.timed)
  void StageBytes(uint64_t out[6]);                // algorithmic bytes per stage of the job prepared last
  int64_t Info(const char* name);                  // "device_bytes", "jobs", "shared_big_bytes", "shared_coef_bytes", "private_plane_jobs", batch infos of the last job ("hf_nonzeros", "lf_simt_frames" ...)
  double prepare_seconds_total() const { return prepare_s_total_; }
  int64_t prepared_jobs() const { return prepared_jobs_; }
  int device() const { return device_; }

 private:
  enum State { kQueued = 0, kPreparing, kFrontIssued, kTailIssued, kHarvested };
  struct Job {
    int64_t ticket = 0;
    std::vector<const uint8_t*> datas; std::vector<size_t> sizes;
    std::vector<void*> device_out, host_out; std::vector<size_t> capacity;
    OutputSpec spec;
    std::vector<int> batch_index;              // per image: index in the batch, -1 = failed before it got there
    PipelineJobResult result;
    std::string job_error;                     // the whole job failed (Prepare threw)
    State state = kQueued;
    bool hf_issued = false, waited = false, cold_wide = false;
    bool wide_chain = false;     // one of the chained cold-start jobs
    bool lf_wide = false;        // ... whose LF stage really is the one-wavefront-per-stream launch (set before wide_enqueued_ reaches its ticket)
    int64_t wide_after = -1;     // cold-start job (one-wavefront-per-stream LF kernel) behind another one: its LF stage waits for that job's (round 6: four at once took 95-120 ms each, one after the other 55)
    void* done_event = nullptr;                // (timing enabled) recorded behind the job's last copy
  };
  struct Slot {
    std::unique_ptr<Batch> batch;
    void *lf_done = nullptr, *front_done = nullptr, *hf_done = nullptr, *idct_done = nullptr, *rest_done = nullptr;
    std::mutex mu;                             // harvest of the slot's current job
  };
  void PrepareWorker(int worker);
  void IssuerLoop();
  void PrepareJob(Job* j, int worker);
  void IssueHf(Job* j);
  void IssueTail(Job* j);
  void Harvest(Job* j);                        // waits for the job's completion on the GPU, fills j->result (once)
  Job* FindJob(int64_t ticket);                // (mu_ held)
  void GrowSharedWhenIdle();                   // (mu_ held, nothing in flight)
  void ReservePlanes(SharedPlanes* sp, size_t bytes);

  const int device_;
  PipelineOptions opt_;
  int nbuf_ = 1, ncoef_ = 2;
  std::vector<std::unique_ptr<Slot>> slots_;
  SharedPlanes big_;
  std::vector<SharedPlanes> coef_;
  size_t want_big_ = 0, want_coef_ = 0;        // the largest layouts seen: what the shared planes grow to when the pipeline is idle
  void* main_ = nullptr; void* d2h_[2] = {nullptr, nullptr};     // [0]: copies to host + status words of every job, in job order ([1] spare: two streams taking turns measured slower)
  std::vector<void*> lf_side_, hf_side_;
  std::vector<void*> tail_side_; std::mutex tail_mu_;   // TailStream()
  void* TailStream(int k);
  void* clock_event_ = nullptr;
  std::mutex mu_;
  std::condition_variable cv_;
  std::map<int64_t, std::shared_ptr<Job>> jobs_;   // submitted, not yet waited for (bounded: old harvested jobs nobody waits for are dropped)
  std::deque<std::shared_ptr<Job>> prep_queue_;
  int64_t next_ticket_ = 0, next_issue_ = 0, completed_upto_ = 0;   // completed_upto_: every ticket below has state >= kHarvested or has been dropped
  int64_t cold_count_ = 0;
  int64_t wide_enqueued_ = -1;    // ticket of the last chained cold-start job whose LF stage has been enqueued (or given up)
  int64_t private_plane_jobs_ = 0;
  bool shutdown_ = false;
  std::vector<std::thread> workers_;
  std::thread issuer_;
  double prepare_s_total_ = 0; int64_t prepared_jobs_ = 0;
  uint64_t last_stage_bytes_[6] = {0, 0, 0, 0, 0, 0};
  std::map<std::string, int64_t> last_info_;
};

// ---- concurrent callers of the libjxl-compatible API (jxl_abi.cc): one shared pipeline per device -----------------------------------------
// A JxlDecoder is used from one thread at a time, different instances concurrently (jpegxl-rs/src/decode.rs:523-532).  Plain one-shot decodes
// (one frame, a caller buffer, no memory manager of the caller's) are handed to the device's scheduler: requests that arrive while earlier ones
// are being prepared ride in one job, so T threads calling decode_with get the batch throughput of the pipeline instead of T serialised decodes.
// Returns 0 and fills dst, or 1 with *error set.
int SchedulerDecode(int device, const uint8_t* data, size_t size, const OutputSpec& spec, void* dst, size_t dst_size, std::string* error);
// statistics of the device's scheduler since the process started: jobs submitted, images decoded (nullptr-safe)
void SchedulerStats(int device, int64_t* jobs, int64_t* images);
void SchedulerNoteDecoder(int delta);   // a JxlDecoder was created (+1) / destroyed (-1): the scheduler sizes its jobs by the decoders that exist (one per calling thread in the reference crate)
void SchedulerShutdown();      // joins the scheduler threads and frees their pipelines (tests; atexit)

}  // namespace jxlhip
