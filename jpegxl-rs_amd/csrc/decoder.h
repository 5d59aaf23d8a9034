// jxl-hip: batch decoder — owns device memory for a batch of frames and enqueues the kernel pipeline.
// A batch of one frame backs the libjxl-compatible JxlDecoder ABI (jxl_abi.cc); larger batches back the resident
// batch API used by bench.py (include/jxl_hip.h).
#pragma once
#include "host_parse.h"
#include "kernels.h"
#include "jpeg_recon.h"
#include <memory>
#include <string>
#include <functional>
#include <vector>

namespace jxlhip {

struct OutputSpec {
  uint32_t num_channels = 0;   // 0 = colour + alpha
  uint32_t type = 0;           // 0 u8, 1 u16, 2 f32, 3 f16
  uint32_t big_endian = 0;
  size_t align = 0;
  void* device_ptr = nullptr;  // optional caller-owned device destination
  bool keep_orientation = false;  // false: the header's orientation is applied to the output (libjxl's default)
  bool unpremul_alpha = false;    // JxlDecoderSetUnpremultiplyAlpha: premultiplied colour is divided by alpha in the write stage
  int upto_frame = -1;            // >= 0: coalesced output of an animation frame — the canvas as it stands after frame `upto_frame` has been blended (the frames behind it are not shown)
  int only_frame = -1;            // >= 0: non-coalesced output (JxlDecoderSetCoalescing(false)) — frame `only_frame` of the image as coded: its own size, its own
                                  // pixels after the colour transform, not blended onto the canvas
  int alpha_from_extra = -1;      // which extra channel fills the alpha slot of 2 / 4-channel output: -1 = the first one of type alpha; >= 0: that extra channel, as it is
                                  // (JxlDecoderSetExtraChannelBuffer hands out any extra channel through this slot)
  uint32_t int_bits = 0;          // integer output: 0 = the full range of the sample type, else samples in [0, 2^int_bits - 1] (JxlDecoderSetImageOutBitDepth)
  bool render_spotcolors = true;  // JxlDecoderSetRenderSpotcolors: spot-colour extra channels are mixed into the colour channels (stage_spot.cc)
};

// What the frames of one image share: the codestream and the image header.
struct ImageShared { Codestream cs; ImageHeader ih; MetadataBoxes boxes; };   // jbrd: payload of the JPEG-reconstruction box, if any

// One *frame* of an image: the unit the decode stages work on (one FrameDev each).  A single-frame image without image
// features is one unit that writes its pixels itself; the frames of other ("complex") images end in float planes and a
// host-planned tail (patches, splines, noise, colour transform, blending, write) composes them (Batch::PlanPostOps).
struct ImageEntry {
  std::shared_ptr<ImageShared> shared;
  Codestream& cs;
  ImageHeader& ih;
  FramePlan plan;
  uint64_t frame_bitpos = 0;
  OutputSpec out;                      // (first unit of the image)
  size_t out_stride = 0, out_size = 0;
  vec<int> deliver_frames;             // (first unit) coalesced animation decoded once: the canvas after each of these frames goes to its own slot of the output area (SetOutputAllFrames)
  bool has_jbrd = false;
  int pub_index = 0, frame_index = 0;  // image the frame belongs to, position among its frames
  bool complex = false;                // frame of a complex image
  uint32_t visible_frame_index = 0, nonvisible_frame_index = 0;   // noise seeds (dec_frame.cc)
  uint64_t preview_bitpos = 0;             // where the preview frame starts (images with a preview: ih.have_preview)
  std::shared_ptr<ImageEntry> lf_source;   // the LF frame (frame type 1) a frame with use_lf_frame takes its LF image from; not a unit of the batch itself
  // arena offsets
  size_t off_cs = 0, off_sec = 0, off_tree = 0, off_bcm = 0;
  size_t off_out = 0;
  explicit ImageEntry(std::shared_ptr<ImageShared> s) : shared(std::move(s)), cs(shared->cs), ih(shared->ih) {}
};

// growable byte buffer in pinned host memory: the constant arena is assembled in it and uploaded from it (decoder.cc)
class HostStage {
 public:
  HostStage() = default;
  HostStage(const HostStage&) = delete;
  HostStage& operator=(const HostStage&) = delete;
  ~HostStage();
  size_t size() const { return n_; }
  uint8_t* data() { return p_; }
  const uint8_t* data() const { return p_; }
  void clear() { n_ = 0; }
  void Resize(size_t n);
  void Reserve(size_t n);      // capacity only
  bool pinned() const { return pinned_; }
 private:
  void Release();
  uint8_t* p_ = nullptr; size_t n_ = 0, cap_ = 0; bool pinned_ = false;
  std::vector<std::pair<uint8_t*, bool>> retired_;     // outgrown blocks, freed with the object
};

// Coefficient / pixel planes that belong to somebody else (a Pipeline): a batch whose layout fits uses them instead of arenas of its own (Batch::UseSharedPlanes).
// `dirty` / `clean_extent` are the coefficient planes' bookkeeping (see Batch::ClearCoefficientsBeforeHf); the owner serialises the decodes that share one set.
struct SharedPlanes { uint8_t* p = nullptr; size_t cap = 0; bool dirty = true; size_t clean_extent = 0; };

struct StageTimes { float lf_ms = 0, lfpost_ms = 0, hf_ms = 0, idct_ms = 0, filter_ms = 0, out_ms = 0, total_ms = 0; };

class Batch {
 public:
  explicit Batch(int device);
  ~Batch();
  // Parses headers (container, image header, frame header, TOC, global sections).  Throws ParseError.
  // allow_partial: a single-frame VarDCT image whose bytes end inside its PassGroup sections is accepted as far as its LF part (LfGlobal, LfGroups, HfGlobal complete):
  // it decodes with every AC coefficient zero — what JxlDecoderFlushImage shows at the kDC progression step
  int AddImage(const uint8_t* data, size_t size, bool allow_partial = false);
  bool is_partial(int i) const;
  // The same for n images, parsed on `threads` host threads and appended in order; returns the index of the first.
  int AddImages(const uint8_t* const* datas, const size_t* sizes, int n, int threads);
  // Same, but an image that does not parse is left out instead of failing the call: (*index)[i] = its index in the batch or -1, (*errors)[i] = what it threw.
  void AddImagesTolerant(const uint8_t* const* datas, const size_t* sizes, int n, int threads, vec<int>* index, std::vector<std::string>* errors);
  // Forgets the images, keeps device arenas / staging buffer / buffer sharing: the object can be filled and prepared again.
  void Reset();
  size_t size() const { return pub_.size(); }
  ImageEntry& image(int i) { return *images_[pub_[i].first_unit]; }
  int num_units() const { return (int)images_.size(); }          // frames of all images, in decode order
  const ImageEntry& unit(int i) const { return *images_[i]; }
  static size_t OutputStride(const ImageHeader& ih, const OutputSpec& o, uint32_t* channels);
  static uint32_t OrientedWidth(const ImageHeader& ih, const OutputSpec& o);
  static uint32_t OrientedHeight(const ImageHeader& ih, const OutputSpec& o);
  static size_t OutputSize(const ImageHeader& ih, const OutputSpec& o);
  // size of the rectangle output `o` of image i covers: the image, or — non-coalesced output — frame o.only_frame as coded (before orientation)
  void OutputDims(int i, const OutputSpec& o, uint32_t* w, uint32_t* h) const;
  size_t OutputSizeOf(int i, const OutputSpec& o) const;
  // Coalesced animation in one decode: the canvas as it stands after every frame of `frames` (ascending positions among the image's frames) is written to slot k of the
  // image's output area in device memory (internal buffers only); CopyOutputSlotToHost hands them out.  Throws "unsupported" for what needs the canvas changed in place on
  // delivery (spot colours, a transfer function deferred behind them): the caller then decodes frame by frame (OutputSpec::upto_frame).
  void SetOutputAllFrames(int i, const OutputSpec& o, const vec<int>& frames);
  void CopyOutputSlotToHost(int i, int slot, void* dst, size_t size, void* stream);
  int num_frames(int i) const { return pub_[i].num_units; }
  const ImageEntry& frame(int i, int k) const { return *images_[pub_[i].first_unit + k]; }
  void SetOutput(int i, const OutputSpec& o);
  // Allocates device memory, uploads streams + tables (inputs become HBM-resident).  stream: hipStream_t.
  // wait_upload = false: returns with the upload still in flight on `stream` (the caller enqueues the first stage of the decode on that same stream)
  void Prepare(void* stream, bool wait_upload = true);
  // Enqueues the whole decode of every frame of the batch.  No host synchronisation inside.
  void Run(void* stream);
  // Waits, checks device status words; throws ParseError on stream errors.
  void Finish(void* stream);
  // JPEG bit-stream reconstruction of image i (decode.rs:493 reconstruct): true if the image carries a usable `jbrd` box and is a plain
  // single-frame JPEG transcode (DCT8 only, RAW quant table, 4:4:4).  ReconstructJpeg runs the entropy-decode stages on the GPU, gathers
  // the quantised coefficients in JPEG layout and serialises the file on the host; throws ParseError if something does not fit.
  bool CanReconstructJpeg(int i, std::string* why = nullptr);
  vec<uint8_t> ReconstructJpeg(int i, void* stream);
  // Copies frame i's pixels to host memory (after Finish).
  void CopyOutputToHost(int i, void* dst, size_t size, void* stream);
  // the preview of image i (JXL_DEC_PREVIEW_IMAGE): its header (the image header with the preview's size), the size of its output, the decode into host memory
  ImageHeader PreviewHeaderOf(int i) const;
  size_t PreviewOutputSize(int i, const OutputSpec& o) const;
  void DecodePreview(int i, const OutputSpec& o, void* dst, size_t cap, void* stream);
  void* device_output(int i) const;
  // Same as Run but brackets every stage with HIP events recorded on `stream` (no host sync).  CollectTimes() waits for
  // all recorded runs and returns the per-stage sums (ms) and the number of runs; used by bench.py for the roofline.
  void RunTimed(void* stream);
  void RunPart(void* stream, int part, bool timed);
  StageTimes CollectTimes(int* runs);
  void DebugTimeline(void* ref_event, float out[9]);
  // ALGORITHMIC bytes per stage for one Run of the batch: 0 lf, 1 lfpost, 2 hf, 3 idct, 4 filters, 5 out
  void StageBytes(uint64_t out[6]) const;
  LaunchCfg cfg;
  size_t const_bytes() const { return const_size_; }
  size_t work_bytes() const { return work_size_ + (coef_owner_ || coef_is_ext_ ? 0 : coeff_bytes_) + (big_owner_ || big_is_ext_ ? 0 : big_size_); }
  void ShareBigArena(Batch* owner);
  void ShareCoefArena(Batch* owner);
  // Planes owned by the caller (pipeline.cc): used by the next Prepare when they are large enough for the batch's layout, otherwise the batch falls back to arenas of
  // its own (big_bytes_wanted / coef_bytes_wanted say what it would have taken).  nullptr = back to own arenas.  The caller orders the decodes that share a set.
  void UseSharedPlanes(SharedPlanes* big, SharedPlanes* coef);
  // Side streams (hipStream_t, owned by the caller, alive as long as the batch decodes) for the independent inverse-transform chains of a batch's Modular images; without them the chains run one after the other.
  void SetTailStreams(std::function<void*(int)> provider) { tail_streams_ = std::move(provider); }
  size_t big_bytes_wanted() const { return big_size_; }
  size_t coef_bytes_wanted() const { return coeff_bytes_; }
  bool uses_shared_big() const { return big_is_ext_; }
  bool uses_shared_coef() const { return coef_is_ext_; }
  // Stream-ordered form of Finish for pipelined callers: EnqueueStatusReadback copies the per-unit status words to pinned host memory behind whatever `stream` holds
  // and records an event; HarvestStatus waits for that event only and returns the words (0 = decoded).  Nothing touches the legacy NULL stream.
  void EnqueueStatusReadback(void* stream);
  void HarvestStatus(vec<uint32_t>* per_unit);
  int first_unit_of(int i) const { return pub_[i].first_unit; }
  int num_units_of(int i) const { return pub_[i].num_units; }
  int device() const { return device_; }
  int64_t Info(const std::string& name) const;
  size_t DebugRead(int i, const std::string& name, int c, void* dst, size_t cap, void* stream);
  uint64_t total_pixels() const;
  uint64_t compressed_bytes() const;

 private:
  void BuildFrameDev(int i, FrameDev* fd, uint8_t* hconst, size_t* const_off, bool measure_only);
  void UploadFrames(void* stream);
  void EnqueueVarDCTFront(void* stream);
  int device_;
  vec<std::unique_ptr<ImageEntry>> images_;   // decode units (frames), in image order
  struct PubImage { int first_unit = 0, num_units = 1; bool complex = false; };
  vec<PubImage> pub_;                          // images as the caller counts them
  // ---- frame tail of complex images
  struct ComplexBufs {      // big-arena offsets of one complex unit ((size_t)-1: not allocated)
    size_t ecf[4], up[3], up_ec[4], noise[3], rgb[3], canvas[3], canvas_ec[4], pa[3], pb[3];
    vec<size_t> ec_int;     // work-arena offsets of the decoded extra channels (int32, coded size)
    size_t color_int[3];            // Modular frames: work-arena offsets of the colour channels after the inverse transforms
    uint32_t nb_color_int = 0;
  };
  vec<ComplexBufs> cbufs_;
  vec<std::function<void(void*)>> post_ops_;
  // ---- LF frames: the frames that units of this batch take their LF image from are decoded by a batch of their own (each as a one-frame image that ends
  // in its XYB planes), run in front of this batch's LF post-processing; its frame tail copies the planes into the referring unit's LF planes
  struct LfTarget { float* dst[3] = {nullptr, nullptr, nullptr}; uint32_t pitch = 0, w = 0, h = 0; };
  std::unique_ptr<Batch> lf_batch_;
  vec<LfTarget> lf_targets_;                   // (of the batch that decodes LF frames: per image, where its planes go)
  vec<std::unique_ptr<JpegData>> jpeg_data_;   // per image, parsed lazily by CanReconstructJpeg
  bool any_complex_ = false;
  void PlanPostOps(HostStage& hconst, const vec<size_t>& up_weights_off);
  void EnqueuePostOps(void* stream);
  vec<FrameDev> frames_host_;
  HostStage hconst_;
  struct ParsedImage { vec<std::unique_ptr<ImageEntry>> units; bool complex = false; };
  static void ParseImage(const uint8_t* data, size_t size, ParsedImage* out, bool allow_partial = false);
  int AddImagesImpl(const uint8_t* const* datas, const size_t* sizes, int n, int threads, bool tolerant, vec<int>* index, std::vector<std::string>* errors);
  int Append(ParsedImage&& im);
  bool DevReserve(void** ptr, size_t* cap, size_t bytes);
  size_t const_cap_ = 0, work_cap_ = 0, big_cap_ = 0, coef_cap_ = 0, coef_laid_out_ = 0, frames_cap_ = 0, passes_cap_ = 0, local_cap_ = 0;
  uint8_t* dconst_ = nullptr; size_t const_size_ = 0;
  uint8_t* dwork_ = nullptr; size_t work_size_ = 0;
  uint8_t* dbig_ = nullptr; size_t big_size_ = 0;   // coefficient + pixel planes (rest half only); may alias big_owner_'s
  Batch* big_owner_ = nullptr; int big_sharers_ = 0;
  // The coefficient planes must be zero when the HF stage starts.  A 7 ms memset per 256 4K frames with nothing else
  // running is avoided like this: every batch has coefficient planes of its own (288 GB of HBM: three batches in flight
  // hold 82 GB of them), and a decode that is done with them clears them again on an internal stream — which the GPU
  // runs under the next batch's latency-bound HF stage.  The next decode of this batch only waits for that event.
  uint8_t* dcoef_ = nullptr;
  void* clear_stream_ = nullptr; void* clear_event_ = nullptr; void* idct_event_ = nullptr;
  vec<void*> mod_join_events_; void* mod_fork_event_ = nullptr;   // fork / join of the Modular tail's side streams (EnqueueModularTail): one inverse-transform chain per image
  std::function<void*(int)> tail_streams_;                        // k -> the owner's k-th side stream for those chains (nullptr: no more)
  bool clear_pending_ = false, coef_dirty_ = true;
  Batch* coef_owner_ = nullptr;
  size_t coef_clean_extent_ = 0;   // bytes of this object's own coefficient planes known to be zero after a completed decode
  SharedPlanes* ext_big_ = nullptr; SharedPlanes* ext_coef_ = nullptr; bool big_is_ext_ = false, coef_is_ext_ = false;
  uint32_t* status_pinned_ = nullptr; size_t status_pinned_n_ = 0; void* status_event_ = nullptr; bool status_pending_ = false;
  bool& CoefDirty() { return coef_is_ext_ ? ext_coef_->dirty : coef_owner_ ? coef_owner_->coef_dirty_ : coef_dirty_; }
  void ClearCoefficientsBeforeHf(void* stream);
  void ClearCoefficientsAfterDecode(void* stream);
  bool has_plane_b_ = false;
  uint32_t* flags_pinned_ = nullptr; size_t flags_pinned_n_ = 0; void* flags_event_ = nullptr; bool flags_pending_ = false;   // placement flags on their way to the host
  void ApplyIdctFlags(const uint32_t* flags);
  void CheckFilterBuffers() const;
  FrameDev* dframes_ = nullptr;
  PassDev* dpasses_ = nullptr;
  vec<PassDev> passes_host_;
  vec<size_t> pass_first_;
  ModLocalDev* dlocal_ = nullptr;      // descriptors of Modular sub-streams with their own tree / code (FrameDev::mod_local)
  vec<ModLocalDev> local_host_;
  vec<size_t> local_first_;
  bool any_multipass_ = false;
  size_t flags_off_ = 0, hfw_off_ = 0;
  vec<uint32_t> hf_written_;   // per unit: non-zero AC coefficients per decode (from the device counter, read by Finish)
  uint32_t decodes_since_finish_ = 0;
  bool ran_once_ = false;         // a complete decode (incl. the LF stage) has been enqueued since Prepare
  size_t coeff_off_ = 0, coeff_bytes_ = 0, status_off_ = 0, modplane_off_ = 0, modplane_bytes_ = 0;
  bool prepared_ = false;
  int max_lf_groups_ = 0, max_groups_ = 0, max_w_ = 0, max_h_ = 0, max_bw_ = 0, max_bh_ = 0, max_epf_ = 0;
  bool any_gab_ = false, any_vardct_ = false, any_modular_ = false;
  FilterPlan fplan_;
  LfSimtPlan lf_simt_;            // SIMT LF decode: device descriptors of the eligible frames' LF-group streams (cfg.lane_stride_lf < 64)
  struct ModFinish { int frame; vec<int> planes; };  // host-side channel lists for modular frames
  vec<vec<size_t>> mod_plane_offsets_;
  // one kernel launch of the host-planned tail of a Modular image: inverse global transforms, then the write stage
  struct ModOp {
    enum Kind { kRct, kPalette, kSqueeze, kOutput } kind = kRct;
    size_t in[4] = {0, 0, 0, 0}, out[4] = {0, 0, 0, 0};   // work-arena offsets
    size_t n = 0;
    uint32_t param = 0, num_c = 0, bits = 0, aw = 0, ah = 0, rw = 0, rh = 0;
    uint32_t nb_deltas = 0, predictor = 0, wp_stride = 0; size_t wp_scratch = 0;   // palette with delta entries / a predictor
    bool has_alpha = false;
    float color_factor = 1.0f, alpha_factor = 1.0f;
    uint32_t float_bits = 0, float_exp_bits = 0;          // float samples: IntToFloatSample instead of the factor
  };
  vec<vec<ModOp>> mod_ops_;
  struct VarDctAlpha { bool has = false; size_t off = 0; float factor = 1.0f; };
  vec<VarDctAlpha> vardct_alpha_;
  bool any_modchan_ = false;      // some frame carries Modular channels (Modular frames, VarDCT frames with extra channels)
  void EnqueueModularTail(void* stream);
  void PlanModularUndo(int i, const std::function<size_t(size_t)>& take);
  vec<vec<void*>> timed_events_;
  size_t timed_rest_cursor_ = 0;       // per frame: work-arena offsets of planes (incl. spare)
};

// device arena pool (decoder.cc): blocks a batch or a pipeline lets go of are kept for the next one that asks for that much memory on that device
size_t DeviceArenaPoolTrim();      // gives every pooled block back to the runtime; returns the bytes released
size_t DeviceArenaPoolHeld();
void* DeviceArenaTake(size_t want, size_t* cap, int device);
void DeviceArenaGive(void* p, size_t cap, int device, bool idle = false);   // idle: nothing on the device refers to the block (no device-wide wait)

}  // namespace jxlhip
