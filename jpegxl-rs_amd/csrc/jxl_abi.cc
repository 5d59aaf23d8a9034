// jxl-hip: libjxl-compatible decoder C ABI (include/jxl_hip.h) on top of the HIP batch decoder.
// Event state machine as exercised by jpegxl-rs/src/decode.rs:207-325 and jpegxl-sys/src/lib.rs:85-171.
#include "../../include/jxl_hip.h"
#include "decoder.h"
#include "pipeline.h"
#include "jpeg_recon.h"
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <new>
#include <memory>
#include <string>

using namespace jxlhip;

static thread_local std::string g_last_error;
static void SetLastError(const std::string& s) { g_last_error = s; }
namespace jxlhip { void SetLastErrorText(const std::string& s) { g_last_error = s; } }     // (gather.cc)

struct JxlDecoderStruct {
  JxlMemoryManager mm;
  bool has_mm;
  MmHooks hooks;        // the same three pointers in the form mm_alloc.h scopes take
  // settings (cleared by Reset)
  int events_wanted;
  bool keep_orientation, unpremul_alpha, render_spotcolors, coalescing;
  float desired_intensity_target;
  JxlParallelRunner runner; void* runner_opaque;
  const uint8_t* input; size_t input_size; bool input_set, input_closed;
  void* out_buffer; size_t out_size; JxlPixelFormat out_format; bool out_set;
  JxlImageOutCallback out_callback; void* out_callback_opaque;   // alternative to out_buffer (rows are handed out after the decode)
  void* preview_buffer; size_t preview_size; JxlPixelFormat preview_format; bool preview_set;   // JxlDecoderSetPreviewOutBuffer
  uint8_t* jpeg_buffer; size_t jpeg_size; bool jpeg_set;
  bool jpeg_available; size_t jpeg_written; vec<uint8_t> jpeg_bytes;   // JPEG bit-stream reconstruction (jbrd)
  // JxlDecoderSetMultithreadedImageOutCallback
  JxlImageOutInitCallback mt_init; JxlImageOutRunCallback mt_run; JxlImageOutDestroyCallback mt_destroy; void* mt_opaque;
  // JxlDecoderSetExtraChannelBuffer (one entry per call, used up by the frame they were set for)
  struct EcBuffer { uint32_t index; void* buffer; size_t size; JxlPixelFormat format; };
  vec<EcBuffer> ec_buffers;
  int progressive_detail;
  uint32_t out_int_bits;     // JxlDecoderSetImageOutBitDepth: 0 = the range of the buffer's sample type
  // box API: the container as handed in (kept only when JXL_DEC_BOX is subscribed), its boxes in file order
  struct BoxRec { char type[4], real[4]; uint64_t raw_size; size_t body, body_size; bool brob; };
  bool decompress_boxes;
  vec<uint8_t> container; vec<BoxRec> boxes;
  size_t box_next, box_split; int box_current; bool box_complete_pending;
  vec<uint8_t> box_plain; bool box_plain_ready; size_t box_written;
  uint8_t* box_buffer; size_t box_size, box_buffer_written; bool box_set; int box_buffer_for;
  // progress
  enum Stage { kInit, kHeaders, kFrame, kDone } stage;
  int events_emitted;
  bool started, need_out_reported;
  // frames as the caller counts them: every regular frame when coalescing is off, the composite (= the last frame) otherwise
  vec<int> frames; size_t frame_cursor, skip_frames; bool frame_announced;
  // coalesced animation decoded once: the canvases after every shown frame wait in device memory, in the format of the first buffer the caller set
  bool anim_cached, anim_cache_failed; JxlPixelFormat anim_format; bool anim_keep_orientation, anim_unpremul, anim_spot; uint32_t anim_int_bits;
  bool partial;        // the batch was parsed from a stream that ends inside its frame's AC groups (Batch::AddImage allow_partial): headers and JxlDecoderFlushImage work, the decode waits for more input
  bool progression_emitted;   // JXL_DEC_FRAME_PROGRESSION (kDC step) has been returned for the current frame
  int progression_passes;     // passes of every group the latest progression step of the current frame shows (0: the kDC step) — what JxlDecoderFlushImage draws
  size_t downsampling_target; // JxlDecoderGetIntendedDownsamplingRatio: 8 at the kDC step, the frame header's ratio for the passes shown after that, 1 for the full image
  bool frame_done;     // JXL_DEC_FULL_IMAGE of frames[frame_cursor] has been returned: its header stays readable until the next JxlDecoderProcessInput moves on
  Batch* batch;
  int device;
  void* stream;        // the decoder's own non-blocking HIP stream (SURVEY 8b "Threading"): created with the first decode, kept over Reset, destroyed with the decoder
};

// One-shot decodes of plain images go to the device's shared pipeline (scheduler.cc): concurrent decoders coalesce into jobs.  JXL_HIP_SCHEDULER=0: every decoder
// runs its own batch of one on its own stream.
static bool SchedulerEnabled() { static const bool on = [] { const char* e = getenv("JXL_HIP_SCHEDULER"); return !(e && e[0] == '0'); }(); return on; }
static void* DecoderStream(JxlDecoder* d) {
  if (!d->stream) {
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); throw ParseError("cannot create a HIP stream for the decoder", false); }
    d->stream = s;
  }
  return d->stream;
}

static int DefaultDevice() {
  const char* e = getenv("JXL_HIP_DEVICE");
  return e ? atoi(e) : 0;
}

// host allocations made on behalf of decoder `d` (parser, tables, staging buffers, the Batch object) go through its memory manager
#define JXL_MM_SCOPE(d) MmScope mm_scope_((d) && (d)->has_mm ? &(d)->hooks : nullptr)
static Batch* NewBatch(int device) { void* mem = MmAllocate(sizeof(Batch)); try { return new (mem) Batch(device); } catch (...) { MmDeallocate(mem); throw; } }
static void DeleteBatch(Batch* b) { if (b) { b->~Batch(); MmDeallocate(b); } }

static void ClearState(JxlDecoder* d) {
  d->events_wanted = 0;
  d->keep_orientation = d->unpremul_alpha = false; d->render_spotcolors = true; d->coalescing = true;
  d->desired_intensity_target = 0;
  d->runner = nullptr; d->runner_opaque = nullptr;
  d->input = nullptr; d->input_size = 0; d->input_set = d->input_closed = false;
  d->out_buffer = nullptr; d->out_size = 0; d->out_set = false; d->out_callback = nullptr; d->out_callback_opaque = nullptr;
  d->preview_buffer = nullptr; d->preview_size = 0; d->preview_set = false;
  d->jpeg_buffer = nullptr; d->jpeg_size = 0; d->jpeg_set = false;
  d->jpeg_available = false; d->jpeg_written = 0; d->jpeg_bytes.clear();
  d->stage = JxlDecoderStruct::kInit; d->events_emitted = 0; d->started = false; d->need_out_reported = false;
  d->frames.clear(); d->frame_cursor = 0; d->skip_frames = 0; d->frame_announced = false; d->frame_done = false;
  d->anim_cached = d->anim_cache_failed = false; d->partial = false; d->progression_emitted = false; d->progression_passes = 0; d->downsampling_target = 1;
  d->mt_init = nullptr; d->mt_run = nullptr; d->mt_destroy = nullptr; d->mt_opaque = nullptr;
  d->ec_buffers.clear(); d->progressive_detail = 1 /* kDC, libjxl's default */; d->out_int_bits = 0;
  d->decompress_boxes = false; d->container.clear(); d->boxes.clear(); d->box_next = d->box_split = 0; d->box_current = -1; d->box_complete_pending = false;
  d->box_plain.clear(); d->box_plain_ready = false; d->box_written = 0; d->box_buffer = nullptr; d->box_size = d->box_buffer_written = 0; d->box_set = false; d->box_buffer_for = -1;
  DeleteBatch(d->batch); d->batch = nullptr;
}
// frames JXL_DEC_FRAME / JXL_DEC_FULL_IMAGE are reported for
static void ListFrames(JxlDecoder* d) {
  d->frames.clear();
  const int n = d->batch->num_frames(0);
  if (d->coalescing) {
    // the composite after the last frame — or, for an animation, after every frame that is shown (frame_header.cc: a regular frame with a duration, or the last one;
    // frames of duration 0 are layers of the frame that follows)
    if (d->batch->image(0).ih.have_animation)
      for (int k = 0; k + 1 < n; k++) { const FramePlan& p = d->batch->frame(0, k).plan; if ((p.frame_type == 0 || p.frame_type == 3) && p.duration > 0) d->frames.push_back(k); }
    d->frames.push_back(n - 1);
    return;
  }
  for (int k = 0; k < n; k++) { const uint32_t t = d->batch->frame(0, k).plan.frame_type; if (t == 0 || t == 3) d->frames.push_back(k); }
}
static int CurrentFrame(const JxlDecoder* d) { return d->batch && d->frame_cursor < d->frames.size() ? d->frames[d->frame_cursor] : -1; }

extern "C" {

const char* JxlHipLastError(void) { return g_last_error.c_str(); }

uint32_t JxlDecoderVersion(void) { return 11002; }  // libjxl 0.11.2 — jpegxl-sys/src/lib.rs:79

JxlSignature JxlSignatureCheck(const uint8_t* buf, size_t len) { return (JxlSignature)CheckSignature(buf, len); }

JxlDecoder* JxlDecoderCreate(const JxlMemoryManager* mm) {
  // The memory manager struct is a temporary on the caller side: copy it (jpegxl-sys decode.rs:394-395).  alloc may
  // unwind (jpegxl-rs/src/memory.rs:140-145): nothing here catches.
  void* mem;
  JxlMemoryManager copy = {nullptr, nullptr, nullptr};
  bool has = false;
  if (mm) {
    if (!!mm->alloc != !!mm->free) return nullptr;
    if (mm->alloc) { copy = *mm; has = true; }
  }
  mem = has ? copy.alloc(copy.opaque, sizeof(JxlDecoderStruct)) : malloc(sizeof(JxlDecoderStruct));
  if (!mem) return nullptr;
  JxlDecoder* d = new (mem) JxlDecoderStruct();
  d->mm = copy; d->has_mm = has; d->batch = nullptr; d->device = DefaultDevice(); d->stream = nullptr;
  d->hooks.opaque = copy.opaque; d->hooks.alloc = copy.alloc; d->hooks.free = copy.free;
  ClearState(d);
  SchedulerNoteDecoder(+1);
  return d;
}
void JxlDecoderReset(JxlDecoder* d) { if (d) ClearState(d); }
void JxlDecoderDestroy(JxlDecoder* d) {
  if (!d) return;
  SchedulerNoteDecoder(-1);
  ClearState(d);
  if (d->stream) { (void)hipStreamSynchronize((hipStream_t)d->stream); (void)hipStreamDestroy((hipStream_t)d->stream); d->stream = nullptr; }
  JxlMemoryManager mm = d->mm; bool has = d->has_mm;
  d->~JxlDecoderStruct();
  if (has) mm.free(mm.opaque, d); else free(d);
}
JxlDecoderStatus JxlDecoderSetParallelRunner(JxlDecoder* d, JxlParallelRunner runner, void* opaque) {
  if (d->started) return JXL_DEC_ERROR;
  d->runner = runner; d->runner_opaque = opaque;  // accepted; the GPU path does not need host worker threads
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSubscribeEvents(JxlDecoder* d, int events) {
  if (d->started) return JXL_DEC_ERROR;
  if (events & 63) return JXL_DEC_ERROR;  // only informative events (>= 0x40) may be subscribed
  d->events_wanted = events;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetKeepOrientation(JxlDecoder* d, JXL_BOOL v) { if (d->started) return JXL_DEC_ERROR; d->keep_orientation = !!v; return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetUnpremultiplyAlpha(JxlDecoder* d, JXL_BOOL v) { if (d->started) return JXL_DEC_ERROR; d->unpremul_alpha = !!v; return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetRenderSpotcolors(JxlDecoder* d, JXL_BOOL v) { if (d->started) return JXL_DEC_ERROR; d->render_spotcolors = !!v; return JXL_DEC_SUCCESS; }
JxlDecoderStatus JxlDecoderSetCoalescing(JxlDecoder* d, JXL_BOOL v) { if (d->started) return JXL_DEC_ERROR; d->coalescing = !!v; return JXL_DEC_SUCCESS; }
void JxlDecoderSkipFrames(JxlDecoder* d, size_t amount) { d->skip_frames += amount; }
JxlDecoderStatus JxlDecoderSkipCurrentFrame(JxlDecoder* d) {
  if (d->stage != JxlDecoderStruct::kFrame || !d->frame_announced || d->frame_done) return JXL_DEC_ERROR;
  d->frame_cursor++; d->frame_announced = false;
  return JXL_DEC_SUCCESS;
}
void JxlDecoderRewind(JxlDecoder* d) {
  JXL_MM_SCOPE(d);
  // (libjxl: "resets the decoder like JxlDecoderReset, but keeps all settings"; the input has to be set again)
  const int events = d->events_wanted; const bool ko = d->keep_orientation, up = d->unpremul_alpha, rs = d->render_spotcolors, co = d->coalescing;
  const float it = d->desired_intensity_target; const JxlParallelRunner runner = d->runner; void* const ro = d->runner_opaque;
  const bool db = d->decompress_boxes; const int pd = d->progressive_detail;
  ClearState(d);
  d->decompress_boxes = db; d->progressive_detail = pd;
  d->events_wanted = events; d->keep_orientation = ko; d->unpremul_alpha = up; d->render_spotcolors = rs; d->coalescing = co;
  d->desired_intensity_target = it; d->runner = runner; d->runner_opaque = ro;
}
static void FillBlendInfo(const BlendInfoH& b, JxlBlendInfo* out) { out->blendmode = (JxlBlendMode)b.mode; out->source = b.source; out->alpha = b.alpha_channel; out->clamp = b.clamp ? 1 : 0; }
JxlDecoderStatus JxlDecoderGetFrameHeader(const JxlDecoder* d, JxlFrameHeader* h) {
  const int k = CurrentFrame(d);
  if (k < 0 || d->stage != JxlDecoderStruct::kFrame || !h) return JXL_DEC_ERROR;
  const ImageEntry& e = d->batch->frame(0, k);
  const FramePlan& p = e.plan;
  memset(h, 0, sizeof(*h));
  h->duration = p.duration; h->timecode = p.timecode; h->name_length = (uint32_t)p.name.size(); h->is_last = p.is_last ? 1 : 0;
  if (d->coalescing) {                     // the composite: no crop, the image's size (jpegxl-sys codestream_header.rs:324-329)
    h->layer_info.xsize = e.ih.xsize; h->layer_info.ysize = e.ih.ysize;
    if (!d->keep_orientation && e.ih.orientation > 4) { h->layer_info.xsize = e.ih.ysize; h->layer_info.ysize = e.ih.xsize; }
    h->is_last = d->frame_cursor + 1 >= d->frames.size() ? 1 : 0;      // (the only composite of a still image; the last shown frame of an animation)
    return JXL_DEC_SUCCESS;
  }
  h->layer_info.have_crop = p.have_crop ? 1 : 0;
  h->layer_info.crop_x0 = p.have_crop ? p.x0 : 0; h->layer_info.crop_y0 = p.have_crop ? p.y0 : 0;
  h->layer_info.xsize = p.frame_w; h->layer_info.ysize = p.frame_h;
  FillBlendInfo(p.blend, &h->layer_info.blend_info);
  h->layer_info.save_as_reference = p.save_as_reference;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetColorAsEncodedProfile(const JxlDecoder* d, JxlColorProfileTarget, JxlColorEncoding* out) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders || !out) return JXL_DEC_ERROR;
  const ImageHeader& ih = d->batch->image(0).ih;
  if (ih.want_icc) { SetLastError("the image carries an ICC profile: no enumerated colour encoding"); return JXL_DEC_ERROR; }
  try {
    memset(out, 0, sizeof(*out));
    out->color_space = ih.color_default ? 0 : (int)ih.color_space;
    out->white_point = ih.color_default ? 1 : (int)ih.white_point;
    out->primaries = ih.color_default || ih.color_space == 1 ? 1 : (int)ih.primaries;
    double w[2], p[6];
    ColorChromaticities(ih, w, p);
    for (int i = 0; i < 2; i++) { out->white_point_xy[i] = w[i]; out->primaries_red_xy[i] = p[i]; out->primaries_green_xy[i] = p[2 + i]; out->primaries_blue_xy[i] = p[4 + i]; }
    if (!ih.color_default && ih.have_gamma) { out->transfer_function = 65535; out->gamma = (double)ih.gamma * 1e-7; }
    else out->transfer_function = ih.color_default ? 13 : (int)ih.tf;
    out->rendering_intent = ih.color_default ? 1 : (int)ih.rendering_intent;
    return JXL_DEC_SUCCESS;
  } catch (const std::exception& e) { SetLastError(e.what()); return JXL_DEC_ERROR; }
}
JxlDecoderStatus JxlDecoderGetExtraChannelInfo(const JxlDecoder* d, size_t index, JxlExtraChannelInfo* out) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders || !out) return JXL_DEC_ERROR;
  const ImageHeader& ih = d->batch->image(0).ih;
  if (index >= ih.extra.size()) return JXL_DEC_ERROR;
  const ExtraChannel& e = ih.extra[index];
  memset(out, 0, sizeof(*out));
  out->type = (int)e.type; out->bits_per_sample = e.depth.bits; out->exponent_bits_per_sample = e.depth.is_float ? e.depth.exp_bits : 0;
  out->dim_shift = e.dim_shift; out->name_length = (uint32_t)e.name.size(); out->alpha_premultiplied = e.alpha_associated ? 1 : 0;
  for (int i = 0; i < 4; i++) out->spot_color[i] = e.spot[i];
  out->cfa_channel = e.cfa_channel;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetExtraChannelName(const JxlDecoder* d, size_t index, char* name, size_t size) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders || !name) return JXL_DEC_ERROR;
  const ImageHeader& ih = d->batch->image(0).ih;
  if (index >= ih.extra.size() || size < ih.extra[index].name.size() + 1) return JXL_DEC_ERROR;
  const vec<char>& n = ih.extra[index].name;
  if (!n.empty()) memcpy(name, n.data(), n.size());
  name[n.size()] = 0;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderSizeHintBasicInfo(const JxlDecoder* d) { return d->batch ? 0 : 98; }      // (decode.cc InitialBasicInfoSizeHint: container signature + box headers + the largest fixed headers)
size_t JxlDecoderGetIntendedDownsamplingRatio(const JxlDecoder* d) { return d ? d->downsampling_target : 1; }      // decode.rs:1495
JxlDecoderStatus JxlDecoderGetFrameName(const JxlDecoder* d, char* name, size_t size) {
  const int k = CurrentFrame(d);
  if (k < 0 || d->stage != JxlDecoderStruct::kFrame || !name) return JXL_DEC_ERROR;
  const vec<char>& n = d->batch->frame(0, k).plan.name;
  if (size < n.size() + 1) return JXL_DEC_ERROR;
  if (!n.empty()) memcpy(name, n.data(), n.size());
  name[n.size()] = 0;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetExtraChannelBlendInfo(const JxlDecoder* d, size_t index, JxlBlendInfo* out) {
  const int k = CurrentFrame(d);
  if (k < 0 || d->stage != JxlDecoderStruct::kFrame || !out) return JXL_DEC_ERROR;
  const FramePlan& p = d->batch->frame(0, k).plan;
  if (index >= p.ec_blend.size()) return JXL_DEC_ERROR;
  FillBlendInfo(p.ec_blend[index], out);
  return JXL_DEC_SUCCESS;
}
// decode.rs:360-362.  libjxl tone-maps (Rec. 2408, then gamut mapping) when the target is below the intensity target of a PQ image, and applies the inverse
// HLG OOTF for the new peak when an HLG image is rendered to another transfer function; everything else is unaffected by the setting.  The tone mapper is not
// built here: the setting is stored, and a decode that WOULD need it fails with a message instead of handing out pixels that were not tone-mapped (NeedsToneMapping).
JxlDecoderStatus JxlDecoderSetDesiredIntensityTarget(JxlDecoder* d, float v) { if (v < 0) return JXL_DEC_ERROR; d->desired_intensity_target = v; return JXL_DEC_SUCCESS; }
static bool NeedsToneMapping(const JxlDecoder* d) {
  if (!(d->desired_intensity_target > 0) || !d->batch) return false;
  const ImageHeader& ih = d->batch->image(0).ih;
  if (!ih.xyb_encoded || ih.want_icc || ih.color_default || ih.have_gamma) return false;    // (only XYB images pass through the stage; sRGB / gamma curves never need it)
  return ih.tf == 16 && d->desired_intensity_target < ih.intensity_target;                   // PQ image towards a dimmer display (HLG images are handed out as HLG: no OOTF change)
}

JxlDecoderStatus JxlDecoderSetInput(JxlDecoder* d, const uint8_t* data, size_t size) {
  if (d->input_set) return JXL_DEC_ERROR;  // libjxl: "already set input, use JxlDecoderReleaseInput first"
  d->input = data; d->input_size = size; d->input_set = true;
  if (d->partial) { d->stage = JxlDecoderStruct::kInit; d->partial = false; }    // (more bytes after a partial parse: parsed again from the start; what has been announced stays announced)
  return JXL_DEC_SUCCESS;
}
void JxlDecoderCloseInput(JxlDecoder* d) { d->input_closed = true; }
size_t JxlDecoderReleaseInput(JxlDecoder* d) {
  // libjxl's streaming sequence: ProcessInput -> NEED_MORE_INPUT, ReleaseInput (= unconsumed bytes), SetInput with more data.
  // Until the headers and the frame index could be parsed nothing counts as consumed (the caller provides the stream again
  // from its start); after that the decoder works from its own copy of the codestream.
  if (!d->input_set) return 0;
  const size_t unconsumed = (d->stage == JxlDecoderStruct::kInit || d->partial) ? d->input_size : 0;     // (a partial parse consumed nothing: the stream comes again from its start, longer)
  d->input = nullptr; d->input_size = 0; d->input_set = false;
  return unconsumed;
}

static void FillBasicInfo(const ImageHeader& ih, JxlBasicInfo* info, bool keep_orientation) {
  memset(info, 0, sizeof(*info));
  info->have_container = ih.have_container;
  // the orientation is applied by the write stage unless the caller keeps it: report the dimensions of what comes out
  const bool transposed = !keep_orientation && ih.orientation > 4;
  info->xsize = transposed ? ih.ysize : ih.xsize; info->ysize = transposed ? ih.xsize : ih.ysize;
  info->bits_per_sample = ih.depth.bits; info->exponent_bits_per_sample = ih.depth.exp_bits;
  info->intensity_target = ih.intensity_target; info->min_nits = ih.min_nits;
  info->relative_to_max_display = ih.relative_to_max_display; info->linear_below = ih.linear_below;
  info->uses_original_profile = !ih.xyb_encoded;
  info->have_preview = ih.have_preview; info->have_animation = ih.have_animation;
  info->preview.xsize = ih.preview_x; info->preview.ysize = ih.preview_y;
  info->orientation = keep_orientation ? (int32_t)ih.orientation : 1;
  info->num_color_channels = ih.color_space == 1 ? 1 : 3;
  info->num_extra_channels = (uint32_t)ih.extra.size();
  for (auto& e : ih.extra) if (e.type == 0) { info->alpha_bits = e.depth.bits; info->alpha_exponent_bits = e.depth.exp_bits; info->alpha_premultiplied = e.alpha_associated; break; }
  info->animation.tps_numerator = ih.tps_num; info->animation.tps_denominator = ih.tps_den; info->animation.num_loops = ih.num_loops;
  info->animation.have_timecodes = ih.have_timecodes;
  info->intrinsic_xsize = ih.intrinsic_x ? ih.intrinsic_x : info->xsize; info->intrinsic_ysize = ih.intrinsic_y ? ih.intrinsic_y : info->ysize;
}

static bool FormatToSpec(const JxlPixelFormat* f, OutputSpec* o) {
  if (!f || f->num_channels > 4) return false;
  switch (f->data_type) {
    case JXL_TYPE_UINT8: o->type = 0; break;
    case JXL_TYPE_UINT16: o->type = 1; break;
    case JXL_TYPE_FLOAT: o->type = 2; break;
    case JXL_TYPE_FLOAT16: o->type = 3; break;
    default: return false;
  }
  o->num_channels = f->num_channels;
  o->big_endian = f->endianness == JXL_BIG_ENDIAN;  // native == little on every supported host
  o->align = f->align;
  return true;
}

JxlDecoderStatus JxlDecoderGetBasicInfo(const JxlDecoder* d, JxlBasicInfo* info) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders) return JXL_DEC_NEED_MORE_INPUT;
  if (info) FillBasicInfo(d->batch->image(0).ih, info, d->keep_orientation);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderImageOutBufferSize(const JxlDecoder* d, const JxlPixelFormat* format, size_t* size) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders) return JXL_DEC_NEED_MORE_INPUT;
  OutputSpec o;
  if (!FormatToSpec(format, &o)) return JXL_DEC_ERROR;
  if (o.num_channels != 0 && o.num_channels < 3 && d->batch->image(0).ih.color_space != 1) { SetLastError("number of channels is too low for colour output"); return JXL_DEC_ERROR; }
  o.keep_orientation = d->keep_orientation;
  o.only_frame = d->coalescing ? -1 : CurrentFrame(d);
  *size = d->batch->OutputSizeOf(0, o);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetImageOutBuffer(JxlDecoder* d, const JxlPixelFormat* format, void* buffer, size_t size) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders) return JXL_DEC_ERROR;
  if (d->out_callback) { SetLastError("an output callback is already set"); return JXL_DEC_ERROR; }
  OutputSpec o;
  if (!FormatToSpec(format, &o)) return JXL_DEC_ERROR;
  if (o.num_channels != 0 && o.num_channels < 3 && d->batch->image(0).ih.color_space != 1) { SetLastError("number of channels is too low for colour output"); return JXL_DEC_ERROR; }
  o.keep_orientation = d->keep_orientation;
  o.only_frame = d->coalescing ? -1 : CurrentFrame(d);
  if (size < d->batch->OutputSizeOf(0, o)) return JXL_DEC_ERROR;
  d->out_buffer = buffer; d->out_size = size; d->out_format = *format; d->out_set = true;
  return JXL_DEC_SUCCESS;
}
// The preview (jpegxl-sys decode.rs:999-1025): announced after the colour encoding when JXL_DEC_PREVIEW_IMAGE is subscribed and the image has one —
// JXL_DEC_NEED_PREVIEW_OUT_BUFFER until a buffer is set, then decoded (a batch of its own for the preview frame) and reported as JXL_DEC_PREVIEW_IMAGE.
static bool PreviewSpec(const JxlDecoder* d, const JxlPixelFormat* format, OutputSpec* o) {
  if (!FormatToSpec(format, o)) return false;
  if (o->num_channels != 0 && o->num_channels < 3 && d->batch->image(0).ih.color_space != 1) { SetLastError("number of channels is too low for colour output"); return false; }
  o->keep_orientation = d->keep_orientation;
  o->unpremul_alpha = d->unpremul_alpha;
  return true;
}
JxlDecoderStatus JxlDecoderPreviewOutBufferSize(const JxlDecoder* d, const JxlPixelFormat* format, size_t* size) {
  if (!d->batch || d->stage < JxlDecoderStruct::kHeaders || !size) return JXL_DEC_ERROR;
  try {
    OutputSpec o;
    if (!d->batch->image(0).ih.have_preview || !PreviewSpec(d, format, &o)) return JXL_DEC_ERROR;
    *size = d->batch->PreviewOutputSize(0, o);
    return JXL_DEC_SUCCESS;
  } catch (const std::exception& e) { SetLastError(e.what()); return JXL_DEC_ERROR; }
}
JxlDecoderStatus JxlDecoderSetPreviewOutBuffer(JxlDecoder* d, const JxlPixelFormat* format, void* buffer, size_t size) {
  size_t need = 0;
  if (!buffer || JxlDecoderPreviewOutBufferSize(d, format, &need) != JXL_DEC_SUCCESS || size < need) return JXL_DEC_ERROR;
  d->preview_buffer = buffer; d->preview_size = size; d->preview_format = *format; d->preview_set = true;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetImageOutCallback(JxlDecoder* d, const JxlPixelFormat* format, JxlImageOutCallback callback, void* opaque) {
  if (!d || !format || !callback || !d->batch || d->stage < JxlDecoderStruct::kHeaders) return JXL_DEC_ERROR;
  if (d->out_set) { SetLastError("an output buffer or callback is already set"); return JXL_DEC_ERROR; }
  size_t need = 0;
  if (JxlDecoderImageOutBufferSize(d, format, &need) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  d->out_format = *format; d->out_buffer = nullptr; d->out_size = need; d->out_set = true;
  d->out_callback = callback; d->out_callback_opaque = opaque;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetJPEGBuffer(JxlDecoder* d, uint8_t* data, size_t size) {
  if (d->jpeg_set) return JXL_DEC_ERROR;
  d->jpeg_buffer = data; d->jpeg_size = size; d->jpeg_set = true;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderReleaseJPEGBuffer(JxlDecoder* d) {
  // jpegxl-sys decode.rs: returns the bytes of the buffer NOT written to.  Like libjxl (JxlToJpegDecoder::WriteOutput) the file is
  // written in one piece: after JXL_DEC_JPEG_NEED_MORE_OUTPUT nothing has been consumed and the whole buffer counts as unused.
  size_t r = d->jpeg_set ? d->jpeg_size - d->jpeg_written : 0;
  d->jpeg_buffer = nullptr; d->jpeg_size = 0; d->jpeg_set = false; d->jpeg_written = 0;
  return r;
}

// ICC profile of the image.  Enumerated colour encodings: synthesised (icc_profile.cc); both targets describe the same encoding,
// the pixels this decoder hands out are always in the codestream's tagged colour space.  Embedded profiles (want_icc): the
// original profile is the decoded one; the pixel data of an XYB image comes out as sRGB (libjxl without a CMS converts to sRGB
// when it cannot target the profile's space), non-XYB samples are in the profile's own space.
static JxlDecoderStatus IccOf(const JxlDecoder* d, JxlColorProfileTarget target, vec<uint8_t>* icc) {
  JXL_MM_SCOPE(d);
  if (!d || !d->batch || d->stage < JxlDecoderStruct::kHeaders) { SetLastError("ICC profile requested before the headers were decoded"); return JXL_DEC_ERROR; }
  const ImageHeader& ih = d->batch->image(0).ih;
  try {
    if (ih.want_icc && (target == JXL_COLOR_PROFILE_TARGET_ORIGINAL || !ih.xyb_encoded)) *icc = ih.icc;
    else if (ih.want_icc) { ImageHeader srgb = ih; srgb.want_icc = false; srgb.icc.clear(); *icc = SynthesizeIcc(srgb); }   // XYB pixel data: rendered to sRGB (grey: same curve)
    else *icc = SynthesizeIcc(ih);
  } catch (const std::exception& e) { SetLastError(e.what()); return JXL_DEC_ERROR; }
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetICCProfileSize(const JxlDecoder* d, JxlColorProfileTarget target, size_t* size) {
  vec<uint8_t> icc;
  if (size) *size = 0;
  if (IccOf(d, target, &icc) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  if (size) *size = icc.size();
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetColorAsICCProfile(const JxlDecoder* d, JxlColorProfileTarget target, uint8_t* out, size_t size) {
  vec<uint8_t> icc;
  if (IccOf(d, target, &icc) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  if (!out || size < icc.size()) { SetLastError("ICC output buffer too small"); return JXL_DEC_ERROR; }
  memcpy(out, icc.data(), icc.size());
  return JXL_DEC_SUCCESS;
}

// ---- decode.rs:1200: rows through init / run / destroy callbacks (one "thread": the caller's, after the image has been decoded on the device)
JxlDecoderStatus JxlDecoderSetMultithreadedImageOutCallback(JxlDecoder* d, const JxlPixelFormat* format, JxlImageOutInitCallback init_cb, JxlImageOutRunCallback run_cb,
                                                            JxlImageOutDestroyCallback destroy_cb, void* init_opaque) {
  if (!d || !format || !init_cb || !run_cb || !destroy_cb || !d->batch || d->stage < JxlDecoderStruct::kHeaders) return JXL_DEC_ERROR;
  if (d->out_set) { SetLastError("an output buffer or callback is already set"); return JXL_DEC_ERROR; }
  size_t need = 0;
  if (JxlDecoderImageOutBufferSize(d, format, &need) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  d->out_format = *format; d->out_buffer = nullptr; d->out_size = need; d->out_set = true;
  d->mt_init = init_cb; d->mt_run = run_cb; d->mt_destroy = destroy_cb; d->mt_opaque = init_opaque;
  return JXL_DEC_SUCCESS;
}

// ---- decode.rs:1224 / :1258: extra channels as planes of their own
static int AlphaIndex(const ImageHeader& ih) { for (size_t i = 0; i < ih.extra.size(); i++) if (ih.extra[i].type == 0) return (int)i; return -1; }
static bool EcSpec(const JxlDecoder* d, const JxlPixelFormat* format, uint32_t index, OutputSpec* o) {
  if (!d || !d->batch || d->stage < JxlDecoderStruct::kHeaders) { SetLastError("extra channel buffers need the basic info"); return false; }
  const ImageHeader& ih = d->batch->image(0).ih;
  if (index >= ih.extra.size()) { SetLastError("no such extra channel"); return false; }
  JxlPixelFormat one = *format; one.num_channels = 1;       // (num_channels of the format is ignored: one sample per pixel)
  if (!FormatToSpec(&one, o)) return false;
  o->keep_orientation = d->keep_orientation;
  o->only_frame = d->coalescing ? -1 : CurrentFrame(d);
  return true;
}
JxlDecoderStatus JxlDecoderExtraChannelBufferSize(const JxlDecoder* d, const JxlPixelFormat* format, size_t* size, uint32_t index) {
  OutputSpec o;
  if (!format || !size || !EcSpec(d, format, index, &o)) return JXL_DEC_ERROR;
  // rows x stride of a one-sample-per-pixel buffer of the (oriented) image size
  uint32_t w = 0, h = 0;
  d->batch->OutputDims(0, o, &w, &h);
  if (!d->keep_orientation && d->batch->image(0).ih.orientation > 4) std::swap(w, h);
  const size_t bps = o.type == 0 ? 1 : o.type == 2 ? 4 : 2;
  size_t stride = (size_t)w * bps;
  if (o.align > 1) stride = (stride + o.align - 1) / o.align * o.align;
  *size = stride * h;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderSetExtraChannelBuffer(JxlDecoder* d, const JxlPixelFormat* format, void* buffer, size_t size, uint32_t index) {
  JXL_MM_SCOPE(d);
  size_t need = 0;
  if (!buffer || JxlDecoderExtraChannelBufferSize(d, format, &need, index) != JXL_DEC_SUCCESS) return JXL_DEC_ERROR;
  if (size < need) { SetLastError("extra channel buffer too small"); return JXL_DEC_ERROR; }
  for (auto& e : d->ec_buffers) if (e.index == index) { e.buffer = buffer; e.size = size; e.format = *format; return JXL_DEC_SUCCESS; }
  d->ec_buffers.push_back(JxlDecoderStruct::EcBuffer{index, buffer, size, *format});
  return JXL_DEC_SUCCESS;
}

// ---- decode.rs:1482 / :1513 / :1528
JxlDecoderStatus JxlDecoderSetProgressiveDetail(JxlDecoder* d, int detail) {
  if (!d || detail < 0 || detail > 3) { SetLastError("unsupported progressive detail (kFrames, kDC, kLastPasses, kPasses are accepted)"); return JXL_DEC_ERROR; }
  // kDC (1) and above: JXL_DEC_FRAME_PROGRESSION is returned when the frame's LF image is decodable — before its AC groups are looked at; kLastPasses (2): again after
  // every pass the frame header names as the last one of a downsampling ratio; kPasses (3): after every pass but the last.  JxlDecoderFlushImage then shows that step
  // (one more decode of the frame on the device with its HF stage cut down to the step's passes), JxlDecoderGetIntendedDownsamplingRatio names its ratio.
  d->progressive_detail = detail;
  return JXL_DEC_SUCCESS;
}
// decode.rs:1513.  Writes the image as far as it can be shown into the buffer set with JxlDecoderSetImageOutBuffer.  After JXL_DEC_FRAME_PROGRESSION: that step — the
// kDC step is the LF image and the HF metadata decoded, every AC coefficient zero; a pass step adds the first passes of every group — through the regular IDCT /
// restoration / colour stages.  After JXL_DEC_NEED_MORE_INPUT on a stream that ends inside its frame's AC groups: every group with the passes that have completely
// arrived, the others from their LF part (dec_frame.cc Flush).  Single-frame VarDCT images without extra channels whose sections come in file order; anything else
// answers JXL_DEC_ERROR ("no flush was done"), as libjxl does when nothing new can be shown.
JxlDecoderStatus JxlDecoderFlushImage(JxlDecoder* d) {
  JXL_MM_SCOPE(d);
  if (!d || !d->batch || d->stage < JxlDecoderStruct::kHeaders || !d->out_set || !d->out_buffer || d->out_callback || d->mt_run) { SetLastError("nothing to flush: no frame in progress or no image out buffer set"); return JXL_DEC_ERROR; }
  if (!d->input_set) { SetLastError("nothing to flush: the input has been released"); return JXL_DEC_ERROR; }
  try {
    if (hipSetDevice(d->device) != hipSuccess) { (void)hipGetLastError(); SetLastError("no usable HIP device"); return JXL_DEC_ERROR; }
    struct Holder { Batch* b; ~Holder() { DeleteBatch(b); } } hold{NewBatch(d->device)};
    hold.b->AddImage(d->input, d->input_size, /*allow_partial=*/true);
    if (hold.b->num_frames(0) != 1) throw ParseError("unsupported: progressive flush of an image with several frames", true);
    const ImageEntry& e = hold.b->frame(0, 0);
    if (e.plan.modular || e.plan.single_section || !e.ih.extra.empty() || e.plan.use_lf_frame || e.complex) throw ParseError("unsupported: progressive flush of this kind of frame", true);
    OutputSpec o;
    FormatToSpec(&d->out_format, &o);
    o.keep_orientation = d->keep_orientation; o.int_bits = d->out_int_bits;
    hold.b->SetOutput(0, o);
    if (hold.b->image(0).out_size > d->out_size) throw ParseError("output buffer too small for this frame", false);
    // what is shown: input that ends inside the frame — every group with the passes that have completely arrived (the HF kernels leave the others out); after a
    // JXL_DEC_FRAME_PROGRESSION event on complete input — that step: the LF image alone (kDC), or the first passes of every group (kLastPasses / kPasses)
    if (!e.plan.partial) {
      if (d->progression_passes <= 0) hold.b->cfg.skip_hf = 1;
      else if ((uint32_t)d->progression_passes < e.plan.num_passes) hold.b->cfg.max_passes = d->progression_passes;
    }
    hold.b->Prepare(DecoderStream(d));
    hold.b->Run(DecoderStream(d));       // ══► the HIP hot path, its HF stage cut down to what the step shows
    hold.b->Finish(DecoderStream(d));
    hold.b->CopyOutputToHost(0, d->out_buffer, hold.b->image(0).out_size, DecoderStream(d));
    return JXL_DEC_SUCCESS;
  } catch (const std::exception& e) {
    SetLastError(std::string("no flush was done: ") + e.what());
    return JXL_DEC_ERROR;
  }
}
JxlDecoderStatus JxlDecoderSetImageOutBitDepth(JxlDecoder* d, const JxlBitDepth* bd) {
  if (!d || !bd || !d->out_set) { SetLastError("JxlDecoderSetImageOutBitDepth: no image out buffer is set"); return JXL_DEC_ERROR; }
  if (bd->type == 0) { d->out_int_bits = 0; return JXL_DEC_SUCCESS; }           // from the pixel format: the full range of the sample type
  if (bd->type != 1 && bd->type != 2) return JXL_DEC_ERROR;
  const uint32_t type_bits = d->out_format.data_type == JXL_TYPE_UINT8 ? 8 : d->out_format.data_type == JXL_TYPE_UINT16 ? 16 : 0;
  const ImageHeader& ih = d->batch->image(0).ih;
  const uint32_t bits = bd->type == 1 ? ih.depth.bits : bd->bits_per_sample;
  const uint32_t exp_bits = bd->type == 1 ? (ih.depth.is_float ? ih.depth.exp_bits : 0) : bd->exponent_bits_per_sample;
  if (type_bits == 0) {                                                         // float output carries no integer range: only a float depth makes sense
    if (bd->type == 1 || exp_bits > 0) { d->out_int_bits = 0; return JXL_DEC_SUCCESS; }
    SetLastError("an integer bit depth for a float buffer"); return JXL_DEC_ERROR;
  }
  if (exp_bits > 0) { SetLastError("a float bit depth for an integer buffer"); return JXL_DEC_ERROR; }
  if (bits == 0 || bits > type_bits) { SetLastError("bit depth does not fit the buffer's sample type"); return JXL_DEC_ERROR; }
  d->out_int_bits = bits == type_bits ? 0 : bits;                               // samples in [0, 2^bits - 1] (the write stage's multiplier)
  return JXL_DEC_SUCCESS;
}

// ---- decode.rs:1326-1470: container boxes
static void ScanBoxesOf(const uint8_t* data, size_t size, vec<JxlDecoderStruct::BoxRec>* boxes, size_t* split) {
  boxes->clear(); *split = 0;
  size_t pos = 0; bool seen_cs = false;
  while (pos + 8 <= size) {
    uint64_t bs = ((uint64_t)data[pos] << 24) | ((uint64_t)data[pos + 1] << 16) | ((uint64_t)data[pos + 2] << 8) | data[pos + 3];
    size_t hdr = 8;
    if (bs == 1) { if (pos + 16 > size) break; bs = 0; for (int i = 0; i < 8; i++) bs = (bs << 8) | data[pos + 8 + i]; hdr = 16; if (bs < 16) break; }
    else if (bs != 0 && bs < 8) break;
    const size_t end = (bs == 0 || bs > (uint64_t)(size - pos)) ? size : pos + (size_t)bs;
    JxlDecoderStruct::BoxRec b;
    memcpy(b.type, data + pos + 4, 4); memcpy(b.real, b.type, 4);
    b.raw_size = bs; b.body = pos + hdr; b.body_size = end - (pos + hdr);
    b.brob = !memcmp(b.type, "brob", 4) && b.body_size >= 4;
    if (b.brob) memcpy(b.real, data + b.body, 4);
    boxes->push_back(b);
    if (!seen_cs && (!memcmp(b.type, "jxlc", 4) || !memcmp(b.type, "jxlp", 4))) { seen_cs = true; *split = boxes->size(); }
    pos = end;
  }
  if (!seen_cs) *split = boxes->size();
}
static void ScanBoxes(JxlDecoder* d) {
  d->box_next = 0; d->box_current = -1;
  ScanBoxesOf(d->container.data(), d->container.size(), &d->boxes, &d->box_split);
}
// content of the current box as the caller gets it: the payload, or — `brob` box with decompression on — the decompressed payload behind the 4-byte type
static bool BoxContent(JxlDecoder* d, const uint8_t** src, size_t* n) {
  const JxlDecoderStruct::BoxRec& b = d->boxes[(size_t)d->box_current];
  if (!(b.brob && d->decompress_boxes)) { *src = d->container.data() + b.body; *n = b.body_size; return true; }
  if (!d->box_plain_ready) {
    if (!BrotliDecompressAll(d->container.data() + b.body + 4, b.body_size - 4, (size_t)1 << 30, &d->box_plain)) { SetLastError("brob box: Brotli stream damaged or libbrotlidec.so.1 not available"); return false; }
    d->box_plain_ready = true;
  }
  *src = d->box_plain.data(); *n = d->box_plain.size();
  return true;
}
// finishes the output of the current box, then announces the next one below `upto`; JXL_DEC_SUCCESS: nothing (more) to report
static JxlDecoderStatus PumpBoxes(JxlDecoder* d, size_t upto) {
  if (!(d->events_wanted & (JXL_DEC_BOX | JXL_DEC_BOX_COMPLETE))) return JXL_DEC_SUCCESS;
  for (;;) {
    if (d->box_current >= 0) {
      if (d->box_set && d->box_buffer_for == d->box_current) {      // (libjxl writes a buffer with the box it was set for only: one that is still set when the next box is announced is left alone)
        const uint8_t* src; size_t n;
        if (!BoxContent(d, &src, &n)) return JXL_DEC_ERROR;
        const size_t k = std::min(n - d->box_written, d->box_size - d->box_buffer_written);
        if (k) memcpy(d->box_buffer + d->box_buffer_written, src + d->box_written, k);
        d->box_written += k; d->box_buffer_written += k;
        if (d->box_written < n) return JXL_DEC_BOX_NEED_MORE_OUTPUT;
      }
      d->box_current = -1; d->box_plain.clear(); d->box_plain_ready = false;
      if (d->events_wanted & JXL_DEC_BOX_COMPLETE) return JXL_DEC_BOX_COMPLETE;
    }
    if (d->box_next >= upto || d->box_next >= d->boxes.size()) return JXL_DEC_SUCCESS;
    d->box_current = (int)d->box_next++; d->box_written = 0;
    if (d->events_wanted & JXL_DEC_BOX) return JXL_DEC_BOX;
  }
}
JxlDecoderStatus JxlDecoderSetBoxBuffer(JxlDecoder* d, uint8_t* data, size_t size) {
  if (!d || d->box_set) { SetLastError("a box buffer is already set: JxlDecoderReleaseBoxBuffer first"); return JXL_DEC_ERROR; }
  if (d->box_current < 0) { SetLastError("no box is current: JxlDecoderSetBoxBuffer follows a JXL_DEC_BOX event"); return JXL_DEC_ERROR; }
  d->box_buffer = data; d->box_size = size; d->box_buffer_written = 0; d->box_set = true; d->box_buffer_for = d->box_current;
  return JXL_DEC_SUCCESS;
}
size_t JxlDecoderReleaseBoxBuffer(JxlDecoder* d) {
  if (!d || !d->box_set) return 0;
  const size_t unused = d->box_size - d->box_buffer_written;
  d->box_buffer = nullptr; d->box_size = d->box_buffer_written = 0; d->box_set = false;
  return unused;
}
JxlDecoderStatus JxlDecoderSetDecompressBoxes(JxlDecoder* d, JXL_BOOL decompress) {
  if (!d) return JXL_DEC_ERROR;
  if (decompress) {
    JXL_MM_SCOPE(d);
    vec<uint8_t> probe;
    const uint8_t empty[1] = {0x06};      // the Brotli stream of an empty output (WBITS 16, ISLAST, ISLASTEMPTY)
    if (!BrotliDecompressAll(empty, 1, 4096, &probe)) { SetLastError("brob boxes cannot be decompressed: libbrotlidec.so.1 not available"); return JXL_DEC_ERROR; }
  }
  d->decompress_boxes = !!decompress;
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetBoxType(JxlDecoder* d, JxlBoxType* type, JXL_BOOL decompressed) {
  if (!d || !type || d->box_current < 0) { SetLastError("no box: the file does not use the container format, or no JXL_DEC_BOX event is current"); return JXL_DEC_ERROR; }
  const JxlDecoderStruct::BoxRec& b = d->boxes[(size_t)d->box_current];
  memcpy(type->type, decompressed ? b.real : b.type, 4);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetBoxSizeRaw(JxlDecoder* d, uint64_t* size) {
  if (!d || !size || d->box_current < 0) return JXL_DEC_ERROR;
  *size = d->boxes[(size_t)d->box_current].raw_size;       // as coded: header included, 0 = "to the end of the file"
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlDecoderGetBoxSizeContents(JxlDecoder* d, uint64_t* size) {
  if (!d || !size || d->box_current < 0) return JXL_DEC_ERROR;
  const JxlDecoderStruct::BoxRec& b = d->boxes[(size_t)d->box_current];
  *size = b.raw_size == 0 ? 0 : b.body_size;               // (libjxl: 0 for a box of unbounded size; a brob box's 4-byte inner type is part of what is handed out undecompressed, so it counts)
  return JXL_DEC_SUCCESS;
}

JxlDecoderStatus JxlDecoderProcessInput(JxlDecoder* d) {
  JXL_MM_SCOPE(d);
  d->started = true;
  // (input is only demanded until the headers have been parsed: the decoder works from its own copy of the codestream afterwards, so a
  // caller may release its buffer — JxlDecoderReleaseInput — and go on)
  if (!d->input_set && d->stage == JxlDecoderStruct::kInit) return d->input_closed ? JXL_DEC_ERROR : JXL_DEC_NEED_MORE_INPUT;
  try {
    if (d->stage == JxlDecoderStruct::kHeaders || d->stage == JxlDecoderStruct::kFrame) {
      // (the current device is per thread, and a decoder may be called from another thread than last time)
      if (hipSetDevice(d->device) != hipSuccess) { (void)hipGetLastError(); SetLastError("no usable HIP device"); return JXL_DEC_ERROR; }
    }
    if (d->stage == JxlDecoderStruct::kInit) {
      JxlSignature sig = JxlSignatureCheck(d->input, d->input_size);
      if (sig == JXL_SIG_INVALID) { SetLastError("invalid signature"); return JXL_DEC_ERROR; }
      if (sig == JXL_SIG_NOT_ENOUGH_BYTES) return d->input_closed ? JXL_DEC_ERROR : JXL_DEC_NEED_MORE_INPUT;
      if (hipSetDevice(d->device) != hipSuccess) { SetLastError("no usable HIP device: the JPEG XL decode path requires an MI355X-class GPU (no CPU fallback)"); return JXL_DEC_ERROR; }
      struct Holder { Batch* b; ~Holder() { DeleteBatch(b); } } hold{NewBatch(d->device)};
      try {
        hold.b->AddImage(d->input, d->input_size);      // (throws "truncated" while the frame index is incomplete)
      } catch (const ParseError& e) {
        // a stream that ends inside its frame's AC groups: the headers are there (the events up to JXL_DEC_NEED_IMAGE_OUT_BUFFER follow), JxlDecoderFlushImage can show the LF
        // part, the decode itself waits for the rest (JXL_DEC_NEED_MORE_INPUT at that point)
        if (strcmp(e.what(), "truncated") != 0) throw;
        DeleteBatch(hold.b); hold.b = nullptr; hold.b = NewBatch(d->device);
        hold.b->AddImage(d->input, d->input_size, /*allow_partial=*/true);     // (throws "truncated" again when not even the LF part is complete)
        d->partial = true;
      }
      DeleteBatch(d->batch);
      d->batch = hold.b; hold.b = nullptr;
      d->stage = JxlDecoderStruct::kHeaders;
      ListFrames(d);
      if ((d->events_wanted & (JXL_DEC_BOX | JXL_DEC_BOX_COMPLETE)) && d->batch->image(0).ih.have_container) {
        d->container.assign(d->input, d->input + d->input_size);
        ScanBoxes(d);
      }
    }
    if (d->stage == JxlDecoderStruct::kHeaders) {
      if (!d->boxes.empty()) { const JxlDecoderStatus bs = PumpBoxes(d, d->box_split); if (bs != JXL_DEC_SUCCESS) return bs; }   // boxes up to the first codestream box
      if ((d->events_wanted & JXL_DEC_BASIC_INFO) && !(d->events_emitted & JXL_DEC_BASIC_INFO)) { d->events_emitted |= JXL_DEC_BASIC_INFO; return JXL_DEC_BASIC_INFO; }
      if ((d->events_wanted & JXL_DEC_COLOR_ENCODING) && !(d->events_emitted & JXL_DEC_COLOR_ENCODING)) { d->events_emitted |= JXL_DEC_COLOR_ENCODING; return JXL_DEC_COLOR_ENCODING; }
      if ((d->events_wanted & JXL_DEC_PREVIEW_IMAGE) && !(d->events_emitted & JXL_DEC_PREVIEW_IMAGE) && d->batch->image(0).ih.have_preview) {
        if (!d->preview_set) return JXL_DEC_NEED_PREVIEW_OUT_BUFFER;
        OutputSpec o;
        if (!PreviewSpec(d, &d->preview_format, &o)) return JXL_DEC_ERROR;
        d->batch->DecodePreview(0, o, d->preview_buffer, d->preview_size, DecoderStream(d));      // ══► the HIP hot path, on the preview frame
        d->events_emitted |= JXL_DEC_PREVIEW_IMAGE;
        return JXL_DEC_PREVIEW_IMAGE;
      }
      // JPEG reconstruction (decode.rs:258-269): announced when the container carries a usable jbrd box; otherwise pixels
      if ((d->events_wanted & JXL_DEC_JPEG_RECONSTRUCTION) && !(d->events_emitted & JXL_DEC_JPEG_RECONSTRUCTION)) {
        d->events_emitted |= JXL_DEC_JPEG_RECONSTRUCTION;
        std::string why;
        if (d->batch->CanReconstructJpeg(0, &why)) { d->jpeg_available = true; return JXL_DEC_JPEG_RECONSTRUCTION; }
        if (d->batch->image(0).has_jbrd) SetLastError("JPEG reconstruction unavailable, decoding to pixels: " + why);
      }
      d->stage = JxlDecoderStruct::kFrame;
    }
    while (d->stage == JxlDecoderStruct::kFrame) {
      // one round per frame the caller sees: the composite (coalescing, one round) or every regular frame as coded
      if (d->frame_done) { d->frame_cursor++; d->frame_announced = false; d->frame_done = false; d->progression_emitted = false; d->progression_passes = 0; d->downsampling_target = 1; }     // (the frame reported last stayed current until now)
      while (d->skip_frames > 0 && d->frame_cursor < d->frames.size() && !d->frame_announced) { d->frame_cursor++; d->skip_frames--; }
      if (d->frame_cursor >= d->frames.size()) { d->stage = JxlDecoderStruct::kDone; break; }
      if ((d->events_wanted & JXL_DEC_FRAME) && !d->frame_announced) { d->frame_announced = true; d->events_emitted |= JXL_DEC_FRAME; return JXL_DEC_FRAME; }
      d->frame_announced = true;
      if (!(d->events_wanted & JXL_DEC_FULL_IMAGE)) { d->frame_cursor++; d->frame_announced = false; continue; }
      if (d->jpeg_available && d->jpeg_set) {
        // the caller asked for the JPEG file: entropy decode on the GPU, Huffman re-encode on the host, written in one piece
        if (d->jpeg_bytes.empty()) {
          try {
            OutputSpec o; o.type = 0; o.num_channels = 3;
            d->batch->SetOutput(0, o);
            d->jpeg_bytes = d->batch->ReconstructJpeg(0, DecoderStream(d));
          } catch (const ParseError& e) {
            if (!e.unsupported) throw;
            SetLastError(std::string(e.what()) + " (decoding to pixels instead)");
            d->jpeg_available = false;
            // the batch was prepared for the JPEG path (output format, buffers): start over for the pixel decode
            if (!d->input_set) throw ParseError("the input was released before the pixel decode could start over", false);
            struct Holder { Batch* b; ~Holder() { DeleteBatch(b); } } hold{NewBatch(d->device)};
            hold.b->AddImage(d->input, d->input_size);
            DeleteBatch(d->batch);
            d->batch = hold.b; hold.b = nullptr;
          }
        }
        if (d->jpeg_available) {
          if (d->jpeg_bytes.size() > d->jpeg_size) return JXL_DEC_JPEG_NEED_MORE_OUTPUT;
          memcpy(d->jpeg_buffer, d->jpeg_bytes.data(), d->jpeg_bytes.size());
          d->jpeg_written = d->jpeg_bytes.size();
          d->stage = JxlDecoderStruct::kDone;
          d->events_emitted |= JXL_DEC_FULL_IMAGE;
          return JXL_DEC_FULL_IMAGE;
        }
      }
      if (!d->out_set) return JXL_DEC_NEED_IMAGE_OUT_BUFFER;
      if (d->partial) {       // the frame's AC groups are not all there yet (JxlDecoderFlushImage shows the LF part meanwhile)
        SetLastError("the stream ends inside the frame");
        return d->input_closed ? JXL_DEC_ERROR : JXL_DEC_NEED_MORE_INPUT;
      }
      if ((d->events_wanted & JXL_DEC_FRAME_PROGRESSION) && d->progressive_detail >= 1 && d->coalescing && d->frames.size() == 1 && d->batch->num_frames(0) == 1) {
        const ImageEntry& e0 = d->batch->frame(0, 0);
        const FramePlan& p0 = e0.plan;
        if (!p0.modular && !p0.single_section && e0.ih.extra.empty() && !p0.use_lf_frame && !e0.complex) {
          // frame_header.h Passes::GetDownsamplingTargetForCompletedPasses
          auto target = [&](uint32_t done) -> size_t {
            if (done >= p0.num_passes) return 1;
            uint32_t r = 8;
            for (uint32_t i = 0; i < p0.num_ds; i++) if (done > p0.ds_last_pass[i]) r = std::min(r, p0.downsample[i]);
            return r;
          };
          if (!d->progression_emitted) { d->progression_emitted = true; d->progression_passes = 0; d->downsampling_target = 8; return JXL_DEC_FRAME_PROGRESSION; }      // the kDC step
          // kLastPasses: a step after every pass the frame header names as the last one of a downsampling ratio; kPasses: after every pass (the last pass is the full image)
          for (uint32_t done = (uint32_t)d->progression_passes + 1; d->progressive_detail >= 2 && done < p0.num_passes; done++) {
            bool step = d->progressive_detail >= 3;
            for (uint32_t i = 0; i < p0.num_ds; i++) step |= p0.ds_last_pass[i] + 1 == done;
            if (step) { d->progression_passes = (int)done; d->downsampling_target = target(done); return JXL_DEC_FRAME_PROGRESSION; }
          }
          d->progression_passes = (int)p0.num_passes; d->downsampling_target = 1;
        }
      }
      if (NeedsToneMapping(d)) throw ParseError("unsupported: desired_intensity_target below the intensity target of a PQ image asks for libjxl's tone mapping stage, which this decoder does not have (leave the target at 0 to get the untouched PQ pixels)", true);
      OutputSpec o;
      FormatToSpec(&d->out_format, &o);
      o.keep_orientation = d->keep_orientation;
      o.unpremul_alpha = d->unpremul_alpha;
      o.render_spotcolors = d->render_spotcolors;
      o.int_bits = d->out_int_bits;
      o.only_frame = d->coalescing ? -1 : d->frames[d->frame_cursor];
      o.upto_frame = d->coalescing && d->frames.size() > 1 ? d->frames[d->frame_cursor] : -1;
      // Animations (coalescing): one decode serves every shown frame — the canvas after each of them is kept in device memory (Batch::SetOutputAllFrames) — as long as
      // the caller keeps asking for the same format and no extra-channel planes; otherwise (and for what that mode cannot do) frame by frame, each replaying the frames before it
      const bool same_format = d->anim_cached && !memcmp(&d->anim_format, &d->out_format, sizeof(JxlPixelFormat)) && d->anim_keep_orientation == d->keep_orientation &&
                               d->anim_unpremul == d->unpremul_alpha && d->anim_spot == d->render_spotcolors && d->anim_int_bits == d->out_int_bits;
      int anim_slot = -1;
      bool scheduled = false;       // the pixels are in the caller's buffer already (shared pipeline)
      if (o.upto_frame >= 0 || (d->coalescing && d->frames.size() > 1)) {
        if (!d->anim_cached && !d->anim_cache_failed && d->ec_buffers.empty()) {
          try {
            OutputSpec oa = o; oa.upto_frame = -1;
            d->batch->SetOutputAllFrames(0, oa, d->frames);
            if (d->batch->image(0).out_size * d->frames.size() > ((size_t)2 << 30)) throw ParseError("unsupported: too many canvases to keep", true);
            if (d->batch->image(0).out_size > d->out_size) { SetLastError("output buffer too small for this frame"); return JXL_DEC_ERROR; }
            d->batch->Prepare(DecoderStream(d));
            d->batch->Run(DecoderStream(d));       // ══► the HIP hot path, once for the whole animation
            d->batch->Finish(DecoderStream(d));
            d->anim_cached = true; d->anim_format = d->out_format; d->anim_keep_orientation = d->keep_orientation; d->anim_unpremul = d->unpremul_alpha; d->anim_spot = d->render_spotcolors; d->anim_int_bits = d->out_int_bits;
            anim_slot = (int)d->frame_cursor;
          } catch (const ParseError& e) {
            if (!e.unsupported) throw;
            d->anim_cache_failed = true;
          }
        } else if (same_format && d->ec_buffers.empty()) anim_slot = (int)d->frame_cursor;
      }
      if (anim_slot < 0) {
        d->anim_cached = false;           // (the batch is prepared for something else from here on)
        d->batch->SetOutput(0, o);
        if (d->batch->image(0).out_size > d->out_size) { SetLastError("output buffer too small for this frame"); return JXL_DEC_ERROR; }
        // A plain one-shot decode into the caller's buffer — what jpegxl-rs' decode_with does (decode.rs:461-484) — rides in a job of the device's shared pipeline:
        // decoders that run on other threads at this moment share the job (scheduler.cc).  Everything else runs as a batch of one on the decoder's own stream.
        const bool via_scheduler = SchedulerEnabled() && !d->has_mm && d->input_set && d->coalescing && d->frames.size() == 1 && o.only_frame < 0 && o.upto_frame < 0 &&
                                   !d->out_callback && !d->mt_run && d->ec_buffers.empty() && d->out_buffer;
        if (via_scheduler) {
          std::string err;
          if (SchedulerDecode(d->device, d->input, d->input_size, o, d->out_buffer, d->batch->image(0).out_size, &err)) throw ParseError(err, err.rfind("unsupported", 0) == 0);   // ══► the HIP hot path
          scheduled = true;
        } else {
          d->batch->Prepare(DecoderStream(d));
          d->batch->Run(DecoderStream(d));       // ══► the HIP hot path
          d->batch->Finish(DecoderStream(d));
        }
      } else if (d->batch->image(0).out_size > d->out_size) { SetLastError("output buffer too small for this frame"); return JXL_DEC_ERROR; }
      auto copy_out = [&](void* dst, size_t size) {
        if (scheduled) return;
        if (anim_slot >= 0) d->batch->CopyOutputSlotToHost(0, anim_slot, dst, size, DecoderStream(d));
        else d->batch->CopyOutputToHost(0, dst, size, DecoderStream(d));
      };
      if (d->out_callback || d->mt_run) {
        // callback output: the image is decoded as a whole on the device, then handed out row by row
        vec<uint8_t> host(d->batch->image(0).out_size);
        copy_out(host.data(), host.size());
        uint32_t w = 0, h = 0;
        d->batch->OutputDims(0, d->batch->image(0).out, &w, &h);
        if (!d->keep_orientation && d->batch->image(0).ih.orientation > 4) std::swap(w, h);
        const size_t stride = d->batch->image(0).out_stride;
        if (d->mt_run) {
          void* run_opaque = d->mt_init(d->mt_opaque, 1, w);
          if (!run_opaque) throw ParseError("the image out init callback failed", false);
          for (size_t y = 0; y < h; y++) d->mt_run(run_opaque, 0, 0, y, w, host.data() + y * stride);
          d->mt_destroy(run_opaque);
        } else for (size_t y = 0; y < h; y++) d->out_callback(d->out_callback_opaque, 0, y, w, host.data() + y * stride);
      } else copy_out(d->out_buffer, d->batch->image(0).out_size);
      if (!d->ec_buffers.empty()) {
        // an extra channel as a plane of its own (JxlDecoderSetExtraChannelBuffer): one more pass of the frame per plane with that channel in the alpha slot of the interleaved
        // output (OutputSpec::alpha_from_extra), in the plane's sample type, de-interleaved here
        for (auto& eb : d->ec_buffers) {
          OutputSpec oe = o;
          JxlPixelFormat one = eb.format; one.num_channels = 1;
          FormatToSpec(&one, &oe);
          const bool grey = d->batch->image(0).ih.color_space == 1;
          oe.num_channels = grey ? 2 : 4; oe.align = 0; oe.device_ptr = nullptr; oe.alpha_from_extra = (int)eb.index; oe.int_bits = 0;
          oe.keep_orientation = o.keep_orientation; oe.unpremul_alpha = false; oe.render_spotcolors = o.render_spotcolors; oe.only_frame = o.only_frame; oe.upto_frame = o.upto_frame;
          d->batch->SetOutput(0, oe);
          d->batch->Prepare(DecoderStream(d)); d->batch->Run(DecoderStream(d)); d->batch->Finish(DecoderStream(d));
          vec<uint8_t> host(d->batch->image(0).out_size);
          d->batch->CopyOutputToHost(0, host.data(), host.size(), DecoderStream(d));
          uint32_t w = 0, h = 0;
          d->batch->OutputDims(0, d->batch->image(0).out, &w, &h);
          if (!d->keep_orientation && d->batch->image(0).ih.orientation > 4) std::swap(w, h);
          const size_t bps = oe.type == 0 ? 1 : oe.type == 2 ? 4 : 2, nc = oe.num_channels, src_stride = d->batch->image(0).out_stride;
          size_t dst_stride = (size_t)w * bps;
          if (eb.format.align > 1) dst_stride = (dst_stride + eb.format.align - 1) / eb.format.align * eb.format.align;
          if (dst_stride * h > eb.size) throw ParseError("extra channel buffer too small for this frame", false);
          for (size_t y = 0; y < h; y++) {
            const uint8_t* sp = host.data() + y * src_stride + (nc - 1) * bps;
            uint8_t* dp = (uint8_t*)eb.buffer + y * dst_stride;
            for (size_t x = 0; x < w; x++) memcpy(dp + x * bps, sp + x * nc * bps, bps);
          }
        }
        d->ec_buffers.clear();
      }
      d->frame_done = true;         // (the cursor moves on with the next call: the frame's header, name and blend info stay readable after JXL_DEC_FULL_IMAGE)
      d->events_emitted |= JXL_DEC_FULL_IMAGE;
      // every layer gets a buffer of its own size; every frame of an animation is asked for anew (decode.cc: the buffer is used up by a frame)
      if (!d->coalescing || d->frame_cursor + 1 < d->frames.size()) { d->out_set = false; d->out_buffer = nullptr; d->out_callback = nullptr; d->mt_run = nullptr; d->out_int_bits = 0; }
      return JXL_DEC_FULL_IMAGE;
    }
    if (!d->boxes.empty()) { const JxlDecoderStatus bs = PumpBoxes(d, d->boxes.size()); if (bs != JXL_DEC_SUCCESS) return bs; }   // the boxes behind the first codestream box
    return JXL_DEC_SUCCESS;
  } catch (const ParseError& e) {
    SetLastError(e.what());
    if (!d->input_closed && !strcmp(e.what(), "truncated")) return JXL_DEC_NEED_MORE_INPUT;
    return JXL_DEC_ERROR;
  } catch (const std::bad_alloc&) {
    SetLastError("out of memory");
    return JXL_DEC_ERROR;
  } catch (const std::exception& e) {     // (length_error from hostile sizes, system_error ...: never unwind into the caller's frames)
    SetLastError(e.what());
    return JXL_DEC_ERROR;
  }
}

// ---- batch extension ---------------------------------------------------------------------------------------------------
struct JxlHipBatchStruct { Batch* b; bool keep_orientation = false; };

JxlHipBatch* JxlHipBatchCreate(int device) {
  try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device >= count) { SetLastError("no usable HIP device (no CPU fallback exists)"); return nullptr; }
    JxlHipBatch* h = new JxlHipBatchStruct();
    h->b = new Batch(device);
    return h;
  } catch (const std::exception& e) { SetLastError(e.what()); return nullptr; }
}
void JxlHipBatchDestroy(JxlHipBatch* h) { if (h) { delete h->b; delete h; } }
int JxlHipBatchAddImage(JxlHipBatch* h, const uint8_t* data, size_t size) {
  try { return h->b->AddImage(data, size); } catch (const std::exception& e) { SetLastError(e.what()); return -1; }
}
int JxlHipBatchAddImages(JxlHipBatch* h, const uint8_t* const* datas, const size_t* sizes, int n, int num_threads) {
  try { return h->b->AddImages(datas, sizes, n, num_threads); } catch (const std::exception& e) { SetLastError(e.what()); return -1; }
}
void JxlHipBatchReset(JxlHipBatch* h) { h->b->Reset(); }
JxlDecoderStatus JxlHipBatchGetBasicInfo(const JxlHipBatch* h, int i, JxlBasicInfo* info) {
  if (i < 0 || (size_t)i >= h->b->size()) return JXL_DEC_ERROR;
  FillBasicInfo(h->b->image(i).ih, info, h->keep_orientation);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlHipBatchOutBufferSize(const JxlHipBatch* h, int i, const JxlPixelFormat* format, size_t* size) {
  OutputSpec o;
  if (i < 0 || (size_t)i >= h->b->size() || !FormatToSpec(format, &o)) return JXL_DEC_ERROR;
  o.keep_orientation = h->keep_orientation;
  *size = Batch::OutputSize(h->b->image(i).ih, o);
  return JXL_DEC_SUCCESS;
}
JxlDecoderStatus JxlHipBatchSetOutput(JxlHipBatch* h, int i, const JxlPixelFormat* format, void* device_buffer) {
  OutputSpec o;
  if (i < 0 || (size_t)i >= h->b->size() || !FormatToSpec(format, &o)) return JXL_DEC_ERROR;
  o.device_ptr = device_buffer;
  o.keep_orientation = h->keep_orientation;
  h->b->SetOutput(i, o);
  return JXL_DEC_SUCCESS;
}
void JxlHipBatchSetLaneStride(JxlHipBatch* h, int lf, int hf) {
  auto ok = [](int v) { return v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
  if (ok(lf)) h->b->cfg.lane_stride_lf = lf;
  if (ok(hf)) h->b->cfg.lane_stride_hf = hf;
}
#define BATCH_TRY(stmt) try { stmt; return JXL_DEC_SUCCESS; } catch (const std::exception& e) { SetLastError(e.what()); return JXL_DEC_ERROR; }
JxlDecoderStatus JxlHipBatchPrepare(JxlHipBatch* h, void* s) { BATCH_TRY(h->b->Prepare(s)) }
JxlDecoderStatus JxlHipBatchDecode(JxlHipBatch* h, void* s) { BATCH_TRY(h->b->Run(s)) }
JxlDecoderStatus JxlHipBatchDecodeTimed(JxlHipBatch* h, void* s) { BATCH_TRY(h->b->RunTimed(s)) }
JxlDecoderStatus JxlHipBatchDecodePart(JxlHipBatch* h, void* s, int part, int timed) {
  if (part < 0 || part > 8) return JXL_DEC_ERROR;
  BATCH_TRY(h->b->RunPart(s, part, timed != 0))
}
JxlDecoderStatus JxlHipBatchCollectTimes(JxlHipBatch* h, JxlHipStageTimes* t, int* runs) {
  BATCH_TRY({ StageTimes st = h->b->CollectTimes(runs); t->lf_ms = st.lf_ms; t->lfpost_ms = st.lfpost_ms; t->hf_ms = st.hf_ms; t->idct_ms = st.idct_ms; t->filter_ms = st.filter_ms; t->out_ms = st.out_ms; t->total_ms = st.total_ms; })
}
JxlDecoderStatus JxlHipBatchFinish(JxlHipBatch* h, void* s) { BATCH_TRY(h->b->Finish(s)) }
void* JxlHipBatchDeviceOutput(const JxlHipBatch* h, int i) { return h->b->device_output(i); }
JxlDecoderStatus JxlHipBatchCopyOutput(JxlHipBatch* h, int i, void* dst, size_t size, void* s) { BATCH_TRY(h->b->CopyOutputToHost(i, dst, size, s)) }
void JxlHipBatchSetOption(JxlHipBatch* h, const char* name, int value) {
  std::string n(name);
  if (n == "force_generic_idct") h->b->cfg.force_generic_idct = value;
  else if (n == "force_unfused_filters") h->b->cfg.force_unfused_filters = value;
  else if (n == "lf_wide_once") h->b->cfg.lf_wide_once = value != 0;
  else if (n == "lf_wp_narrow_test") h->b->cfg.lf_wp_narrow_test = value != 0;
  else if (n == "debug_stop_after" && value >= 0 && value <= 5) h->b->cfg.debug_stop_after = value;
  else if (n == "keep_orientation") h->keep_orientation = value != 0;   // applies to outputs set afterwards
  else if (n == "hf_block_threads" && value >= 64 && value <= 1024 && value % 64 == 0) h->b->cfg.hf_block_threads = value;
  else if (n == "lds_code_budget" && value >= 0 && value <= 128 * 1024) h->b->cfg.lds_code_budget = value;
  else if (n == "lf_force_big" && value >= -1 && value <= 2) h->b->cfg.lf_force_big = value;
  else if (n == "hf_lanes_per_wave" && value >= 0 && value <= 64) h->b->cfg.hf_lanes_per_wave = value;   // SIMT HF stage: group streams per wavefront (1: the wave-wide kernel where it applies); 0: the throughput packing
}
size_t JxlHipBatchDebugRead(JxlHipBatch* h, int index, const char* name, int channel, void* dst, size_t cap, void* s) {
  try { return h->b->DebugRead(index, name ? name : "", channel, dst, cap, s); } catch (const std::exception& e) { SetLastError(e.what()); return 0; }
}
int64_t JxlHipBatchGetInfo(const JxlHipBatch* h, const char* name) {
  try { return h->b->Info(name ? name : ""); } catch (const std::exception& e) { SetLastError(e.what()); return -1; }
}
uint64_t JxlHipBatchTotalPixels(const JxlHipBatch* h) { return h->b->total_pixels(); }
uint64_t JxlHipBatchCompressedBytes(const JxlHipBatch* h) { return h->b->compressed_bytes(); }
void JxlHipBatchStageBytes(const JxlHipBatch* h, uint64_t out[6]) { h->b->StageBytes(out); }
uint64_t JxlHipBatchDeviceBytes(const JxlHipBatch* h) { return h->b->const_bytes() + h->b->work_bytes(); }
int JxlHipBatchShareCoefficients(JxlHipBatch* h, JxlHipBatch* owner) {
  try { h->b->ShareCoefArena(owner ? owner->b : nullptr); return 0; } catch (const std::exception& e) { SetLastError(e.what()); return 1; }
}
int JxlHipBatchShareBuffers(JxlHipBatch* h, JxlHipBatch* owner) {
  try { h->b->ShareBigArena(owner ? owner->b : nullptr); return 0; } catch (const std::exception& e) { SetLastError(e.what()); return 1; }
}

// ---- streaming pipeline (pipeline.h) -----------------------------------------------------------------------------------------
struct JxlHipPipelineStruct { Pipeline* p; };

JxlHipPipeline* JxlHipPipelineCreate(int device, const JxlHipPipelineOptions* o) {
  try {
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) { (void)hipGetLastError(); SetLastError("no usable HIP device (no CPU fallback exists)"); return nullptr; }
    PipelineOptions po;
    if (o) {
      auto pick = [](int32_t v, int def) { return v > 0 ? (int)v : def; };
      po.in_flight = pick(o->jobs_in_flight, po.in_flight); po.lf_streams = pick(o->lf_streams, po.lf_streams); po.hf_streams = pick(o->hf_streams, po.hf_streams);
      po.prepare_threads = pick(o->prepare_threads, po.prepare_threads); po.parse_threads = pick(o->parse_threads, po.parse_threads);
      po.lane_stride_lf = pick(o->lane_stride_lf, po.lane_stride_lf); po.lane_stride_hf = pick(o->lane_stride_hf, po.lane_stride_hf);
      po.wide_first = o->wide_first >= 0 ? o->wide_first : po.wide_first; po.small_job_frames = o->small_job_frames > 0 ? o->small_job_frames : 0;
      po.timed = o->timed != 0;
      po.reserve_frames = o->reserve_frames; po.reserve_width = o->reserve_width; po.reserve_height = o->reserve_height; po.reserve_plane_sets = o->reserve_plane_sets > 1 ? 2 : 1;
    }
    auto ok = [](int v) { return v == 1 || v == 2 || v == 4 || v == 8 || v == 16 || v == 32 || v == 64; };
    if (!ok(po.lane_stride_lf) || !ok(po.lane_stride_hf)) { SetLastError("lane strides must be powers of two up to 64"); return nullptr; }
    JxlHipPipeline* h = new JxlHipPipelineStruct();
    try { h->p = new Pipeline(device, po); } catch (...) { delete h; throw; }
    return h;
  } catch (const std::exception& e) { SetLastError(e.what()); return nullptr; }
}
void JxlHipPipelineDestroy(JxlHipPipeline* h) { if (h) { delete h->p; delete h; } }
int64_t JxlHipPipelineSubmit(JxlHipPipeline* h, const uint8_t* const* datas, const size_t* sizes, int n, const JxlPixelFormat* format, void* const* device_out, void* const* host_out,
                             const size_t* out_capacity) {
  try {
    OutputSpec o;
    if (!h || !FormatToSpec(format, &o)) { SetLastError("JxlHipPipelineSubmit: bad pixel format"); return -1; }
    return h->p->Submit(datas, sizes, n, o, device_out, host_out, out_capacity);
  } catch (const std::exception& e) { SetLastError(e.what()); return -1; }
}
JxlDecoderStatus JxlHipPipelineWait(JxlHipPipeline* h, int64_t ticket, int* image_status, int n, float* end_ms) {
  try {
    PipelineJobResult r;
    h->p->Wait(ticket, &r);
    bool all_ok = true;
    std::string first;
    for (size_t i = 0; i < r.status.size(); i++) {
      if (image_status && (int)i < n) image_status[i] = r.status[i];
      if (r.status[i] != 0) { all_ok = false; if (first.empty()) first = "image " + std::to_string(i) + ": " + r.error[i]; }
    }
    if (end_ms) *end_ms = r.end_ms;
    if (!all_ok) SetLastError(first);
    return all_ok ? JXL_DEC_SUCCESS : JXL_DEC_ERROR;
  } catch (const std::exception& e) { SetLastError(e.what()); if (image_status) for (int i = 0; i < n; i++) image_status[i] = 1; return JXL_DEC_ERROR; }
}
JxlDecoderStatus JxlHipPipelineWaitAll(JxlHipPipeline* h) { BATCH_TRY(h->p->WaitAll()) }
JxlDecoderStatus JxlHipPipelineResetClock(JxlHipPipeline* h) { BATCH_TRY(h->p->ResetClock()) }
JxlDecoderStatus JxlHipPipelineCollectTimes(JxlHipPipeline* h, JxlHipStageTimes* t, int* runs) {
  BATCH_TRY({ StageTimes st = h->p->CollectTimes(runs); t->lf_ms = st.lf_ms; t->lfpost_ms = st.lfpost_ms; t->hf_ms = st.hf_ms; t->idct_ms = st.idct_ms; t->filter_ms = st.filter_ms; t->out_ms = st.out_ms; t->total_ms = st.total_ms; })
}
void JxlHipPipelineStageBytes(JxlHipPipeline* h, uint64_t out[6]) { h->p->StageBytes(out); }
int64_t JxlHipPipelineGetInfo(JxlHipPipeline* h, const char* name) {
  try {
    const std::string n(name ? name : "");
    if (n == "prepare_us_total") return (int64_t)(h->p->prepare_seconds_total() * 1e6);
    if (n == "prepared_jobs") return h->p->prepared_jobs();
    return h->p->Info(name);
  } catch (const std::exception& e) { SetLastError(e.what()); return -1; }
}
void* JxlHipHostAlloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); SetLastError("hipHostMalloc failed"); return nullptr; }
  return p;
}
void JxlHipHostFree(void* p) { if (p) (void)hipHostFree(p); }
size_t JxlHipArenaPoolTrim(void) { return DeviceArenaPoolTrim(); }
size_t JxlHipArenaPoolHeld(void) { return DeviceArenaPoolHeld(); }
void JxlHipSchedulerStats(int device, int64_t* jobs, int64_t* images) { SchedulerStats(device, jobs, images); }
void JxlHipSchedulerShutdown(void) { SchedulerShutdown(); }
JxlDecoderStatus JxlHipImageOutSize(const uint8_t* data, size_t size, const JxlPixelFormat* format, JxlBasicInfo* info, size_t* out_size) {
  try {
    OutputSpec o;
    if (!FormatToSpec(format, &o)) { SetLastError("bad pixel format"); return JXL_DEC_ERROR; }
    std::shared_ptr<ImageShared> sh(new ImageShared());
    bool have_container = false, has_jbrd = false;
    if (!ExtractCodestream(data, size, &sh->cs, &have_container, &has_jbrd, nullptr)) return JXL_DEC_NEED_MORE_INPUT;
    uint64_t bitpos = 0;
    ParseImageHeader(sh->cs, &sh->ih, &bitpos);
    sh->ih.have_container = have_container;
    if (info) FillBasicInfo(sh->ih, info, false);
    if (out_size) *out_size = Batch::OutputSize(sh->ih, o);
    return JXL_DEC_SUCCESS;
  } catch (const std::exception& e) { SetLastError(e.what()); return JXL_DEC_ERROR; }
}

// the pipelines keep ~15 HIP streams busy at once; the runtime maps streams onto 4 hardware queues by default and kernels of streams that share a queue serialise.
// Set (unless the caller chose a value) before the HIP runtime reads it, i.e. when this library is loaded — the runtime reads the variable once, when it initialises, so
// the first pipeline would be too late.  A process-wide side effect of loading the library: JXL_HIP_KEEP_HW_QUEUES=1 leaves the variable alone.
__attribute__((constructor)) static void JxlHipDefaultHwQueues() { if (!getenv("JXL_HIP_KEEP_HW_QUEUES")) setenv("GPU_MAX_HW_QUEUES", "16", 0); }

int JxlHipDebugWriteJpegSampled(const uint8_t* jbrd, size_t jbrd_size, uint32_t width, uint32_t height, const uint32_t* sampling, const int16_t* coefficients,
                                const int32_t* quant_tables, uint8_t* out, size_t* out_size) {
  try {
    JpegData jd;
    std::string err;
    if (!ParseJbrd(jbrd, jbrd_size, &jd, &err)) { SetLastError(err); return 1; }
    const size_t nc = jd.components.size();
    uint32_t max_h = 1, max_v = 1;
    for (size_t c = 0; c < nc; c++) {
      jd.components[c].h_samp = sampling ? sampling[2 * c] : 1; jd.components[c].v_samp = sampling ? sampling[2 * c + 1] : 1;
      if (jd.components[c].h_samp < 1 || jd.components[c].h_samp > 4 || jd.components[c].v_samp < 1 || jd.components[c].v_samp > 4) { SetLastError("bad sampling factor"); return 1; }
      max_h = std::max(max_h, jd.components[c].h_samp); max_v = std::max(max_v, jd.components[c].v_samp);
    }
    const size_t mcu_cols = (width + 8 * max_h - 1) / (8 * max_h), mcu_rows = (height + 8 * max_v - 1) / (8 * max_v);
    const int16_t* planes[3] = {coefficients, coefficients, coefficients};
    size_t first = 0;
    for (size_t c = 0; c < nc; c++) {
      for (int k = 0; k < 64; k++) jd.quant[jd.components[c].quant_idx].values[k] = quant_tables[c * 64 + k];
      planes[c] = coefficients + first * 64;
      first += mcu_cols * jd.components[c].h_samp * mcu_rows * jd.components[c].v_samp;
    }
    vec<uint8_t> bytes;
    if (!WriteJpeg(jd, width, height, planes, &bytes, &err)) { SetLastError(err); return 1; }
    const size_t cap = *out_size;
    *out_size = bytes.size();
    if (cap < bytes.size()) { SetLastError("output buffer too small"); return 2; }
    memcpy(out, bytes.data(), bytes.size());
    return 0;
  } catch (const std::exception& e) { SetLastError(e.what()); return 1; }
}
int JxlHipDebugWriteJpeg(const uint8_t* jbrd, size_t jbrd_size, uint32_t width, uint32_t height, const int16_t* coefficients, const int32_t* quant_tables,
                         uint8_t* out, size_t* out_size) {
  return JxlHipDebugWriteJpegSampled(jbrd, jbrd_size, width, height, nullptr, coefficients, quant_tables, out, out_size);
}

// Host-only (no GPU): parses the container, the image header and every frame header / TOC / LfGlobal / local Modular stream the host
// side handles, and writes a one-line-per-frame description into `out` (NUL-terminated, truncated to cap).  Returns 0, or 1 with
// JxlHipLastError() set when the file is rejected — the host half of AddImage, for tests and triage without a device.
int JxlHipDebugDescribe(const uint8_t* data, size_t size, char* out, size_t cap) {
  try {
    Batch b(-1);
    std::string s;
    try { b.AddImage(data, size); }
    catch (const ParseError& e) {
      // a stream that ends inside its frame's AC groups is still described as far as a progressive flush could show it (Batch::AddImage allow_partial)
      if (strcmp(e.what(), "truncated") != 0) throw;
      b.AddImage(data, size, /*allow_partial=*/true);
      s += "partial: the stream ends inside the frame's AC groups\n";
    }
    char line[512];
    const ImageHeader& ih = b.image(0).ih;
    snprintf(line, sizeof line, "image %ux%u bits=%u extra=%zu xyb=%d gray=%d icc=%zu frames=%d\n", ih.xsize, ih.ysize, ih.depth.bits, ih.extra.size(), (int)ih.xyb_encoded,
             (int)(ih.color_space == 1), ih.icc.size(), b.num_units());
    s += line;
    for (int i = 0; i < b.num_units(); i++) {
      const FramePlan& p = b.unit(i).plan;
      snprintf(line, sizeof line, "frame %d %s type=%u %ux%u at (%d,%d) groups=%u lf_groups=%u passes=%u upsampling=%u patches=%zu splines=%zu noise=%d blend=%u last=%d "
               "tree_nodes=%zu max_prop=%d wp=%d prefix=%d lz77=%d local_streams=%zu transforms=%zu sections=%zu\n", i, p.modular ? "modular" : "vardct", p.frame_type, p.width, p.height,
               p.x0, p.y0, p.num_groups, p.num_lf_groups, p.num_passes, p.upsampling, p.feat.patches.size(), p.feat.splines.size(), (int)p.feat.has_noise, p.blend.mode, (int)p.is_last,
               p.tree.nodes.size(), p.max_prop, (int)p.tree.uses_wp, (int)p.tree_code.use_prefix, (int)p.tree_code.lz77, p.local_streams.size(), p.gtransforms.size(), p.sections.size());
      s += line;
      if (!p.modular) {
        snprintf(line, sizeof line, "  quantizer global_scale=%u quant_lf=%u m_lf=%g,%g,%g x_qm=%u b_qm=%u cfl_base=%g,%g colour_factor=%u ycbcr=%d sampling=%u,%u,%u flags=%llu\n", p.global_scale, p.quant_lf,
                 p.m_lf[0], p.m_lf[1], p.m_lf[2], p.x_qm_scale, p.b_qm_scale, p.base_x, p.base_b, p.color_factor, (int)p.do_ycbcr, p.jpeg_upsampling[0], p.jpeg_upsampling[1], p.jpeg_upsampling[2],
                 (unsigned long long)p.flags);
        s += line;
        for (int k = 0; k < 17; k++) if (p.qspec[k].mode != 0) { snprintf(line, sizeof line, "  qtable kind=%d mode=%u raw_den=%g\n", k, p.qspec[k].mode, p.qspec[k].raw_den); s += line; }
        snprintf(line, sizeof line, "  lf_code contexts=%u clusters=%u log_alpha=%u alias_bytes=%zu\n", p.tree_code.num_ctx, p.tree_code.num_clusters, p.tree_code.log_alpha, p.tree_code.alias.size() * 8);
        s += line;
        for (size_t ps = 0; ps < p.ac_code.size(); ps++) {
          snprintf(line, sizeof line, "  ac_code pass=%zu contexts=%u clusters=%u log_alpha=%u alias_bytes=%zu\n", ps, p.ac_code[ps].num_ctx, p.ac_code[ps].num_clusters, p.ac_code[ps].log_alpha, p.ac_code[ps].alias.size() * 8);
          s += line;
        }
      }
    }
    if (b.image(0).has_jbrd) {
      // the host half of the JPEG-reconstruction check: jbrd parses, the ICC / Exif / XMP markers find their payloads
      std::string why;
      const bool ok = b.CanReconstructJpeg(0, &why);
      s += ok ? std::string("jpeg_reconstruction=1\n") : "jpeg_reconstruction=0 reason: " + why + "\n";
    }
    if (b.image(0).ih.have_container) {
      // the box walk of the decoder's box API (ScanBoxes), on the same bytes
      vec<JxlDecoderStruct::BoxRec> boxes; size_t split = 0;
      ScanBoxesOf(data, size, &boxes, &split);
      snprintf(line, sizeof line, "boxes=%zu before_codestream=%zu:", boxes.size(), split);
      s += line;
      for (auto& bx : boxes) {
        char t[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < 4; k++) t[k] = (bx.type[k] >= 0x20 && bx.type[k] < 0x7F) ? bx.type[k] : '?';      // (damaged files: keep the text printable)
        snprintf(line, sizeof line, " %s(%zu)", t, bx.body_size); s += line;
      }
      s += "\n";
    }
    if (out && cap) { const size_t n = std::min(cap - 1, s.size()); memcpy(out, s.data(), n); out[n] = 0; }
    return 0;
  } catch (const std::exception& e) { SetLastError(e.what()); return 1; }
}

size_t JxlHipLibraryQuantTable(int kind, int c, float* out, size_t cap) {
  try {
    if (kind < 0 || kind >= 17 || c < 0 || c >= 3) return 0;
    vec<float> t;
    ComputeQuantTable(QuantTableSpec(), kind, c, &t);
    for (size_t i = 0; i < t.size() && i < cap; i++) out[i] = t[i];
    return t.size();
  } catch (const std::exception& e) { SetLastError(e.what()); return 0; }
}

int JxlHipColorProfileFromHeaders(const uint8_t* data, size_t size, uint8_t* icc_out, size_t* icc_size) {
  try {
    Codestream cs; bool container = false, jbrd = false;
    if (!ExtractCodestream(data, size, &cs, &container, &jbrd)) { SetLastError("truncated input"); return 1; }
    ImageHeader ih; uint64_t frame_bitpos = 0;
    ParseImageHeader(cs, &ih, &frame_bitpos);
    const vec<uint8_t> icc = ih.want_icc ? ih.icc : SynthesizeIcc(ih);   // (embedded profile: the original one)
    const size_t cap = icc_size ? *icc_size : 0;
    if (icc_size) *icc_size = icc.size();
    if (icc_out) { if (cap < icc.size()) { SetLastError("ICC output buffer too small"); return 1; } memcpy(icc_out, icc.data(), icc.size()); }
    return 0;
  } catch (const std::exception& e) { SetLastError(e.what()); return 1; }
}

}  // extern "C"
