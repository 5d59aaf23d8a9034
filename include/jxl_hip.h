/* jxl_hip.h — C ABI of the MI355X-native JPEG XL decode path.
 *
 * This is the drop-in boundary: the symbols below are exactly the libjxl / libjxl_threads entry points that
 * inflation/jpegxl-rs binds through jpegxl-sys (extern "C-unwind"), with identical names, argument meaning, struct
 * layouts and status codes, exported from libjxl.so / libjxl_threads.so look-alikes so that
 * `DEP_JXL_LIB=<dir> cargo build -p jpegxl-rs` links against this implementation with zero Rust changes
 * (jpegxl-sys/build.rs:30-34).  Each declaration cites the reference line it replaces.
 * The second part (JxlHip*) is an extension without reference counterpart: a device-resident batch decode used by
 * bench.py and by batch users — the reference decodes batches by looping decode_with (jpegxl-rs/benches/decode.rs:16-19).
 */
#ifndef JXL_HIP_H_
#define JXL_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- common types (jpegxl-sys/src/common/types.rs:25-148) -------------------------------------------------------- */
typedef int JXL_BOOL;                                                     /* types.rs:25-31 */
typedef enum { JXL_TYPE_FLOAT = 0, JXL_TYPE_UINT8 = 2, JXL_TYPE_UINT16 = 3, JXL_TYPE_FLOAT16 = 5 } JxlDataType;   /* types.rs:40-55 */
typedef enum { JXL_NATIVE_ENDIAN = 0, JXL_LITTLE_ENDIAN = 1, JXL_BIG_ENDIAN = 2 } JxlEndianness;                  /* types.rs:60-72 */
typedef struct { uint32_t num_channels; JxlDataType data_type; JxlEndianness endianness; size_t align; } JxlPixelFormat; /* types.rs:83-103 */

/* jpegxl-sys/src/common/memory_manager.rs:22-64 */
typedef void* (*jpegxl_alloc_func)(void* opaque, size_t size);
typedef void (*jpegxl_free_func)(void* opaque, void* address);
typedef struct { void* opaque; jpegxl_alloc_func alloc; jpegxl_free_func free; } JxlMemoryManager;

/* jpegxl-sys/src/metadata/codestream_header.rs:38-241 (204 bytes) */
typedef enum { JXL_ORIENT_IDENTITY = 1 } JxlOrientation;
typedef struct { uint32_t xsize, ysize; } JxlPreviewHeader;
typedef struct { uint32_t tps_numerator, tps_denominator, num_loops; JXL_BOOL have_timecodes; } JxlAnimationHeader;
typedef struct {
  JXL_BOOL have_container;
  uint32_t xsize, ysize, bits_per_sample, exponent_bits_per_sample;
  float intensity_target, min_nits;
  JXL_BOOL relative_to_max_display;
  float linear_below;
  JXL_BOOL uses_original_profile, have_preview, have_animation;
  int32_t orientation;
  uint32_t num_color_channels, num_extra_channels, alpha_bits, alpha_exponent_bits;
  JXL_BOOL alpha_premultiplied;
  JxlPreviewHeader preview;
  JxlAnimationHeader animation;
  uint32_t intrinsic_xsize, intrinsic_ysize;
  uint8_t padding[100];
} JxlBasicInfo;

/* jpegxl-sys/src/decode.rs:43-54, 84-247, 284-287 */
typedef enum { JXL_SIG_NOT_ENOUGH_BYTES = 0, JXL_SIG_INVALID = 1, JXL_SIG_CODESTREAM = 2, JXL_SIG_CONTAINER = 3 } JxlSignature;
typedef enum {
  JXL_DEC_SUCCESS = 0, JXL_DEC_ERROR = 1, JXL_DEC_NEED_MORE_INPUT = 2, JXL_DEC_NEED_PREVIEW_OUT_BUFFER = 3,
  JXL_DEC_NEED_IMAGE_OUT_BUFFER = 5, JXL_DEC_JPEG_NEED_MORE_OUTPUT = 6, JXL_DEC_BOX_NEED_MORE_OUTPUT = 7,
  JXL_DEC_BASIC_INFO = 0x40, JXL_DEC_COLOR_ENCODING = 0x100, JXL_DEC_PREVIEW_IMAGE = 0x200, JXL_DEC_FRAME = 0x400,
  JXL_DEC_FULL_IMAGE = 0x1000, JXL_DEC_JPEG_RECONSTRUCTION = 0x2000, JXL_DEC_BOX = 0x4000, JXL_DEC_FRAME_PROGRESSION = 0x8000,
  JXL_DEC_BOX_COMPLETE = 0x10000
} JxlDecoderStatus;
typedef enum { JXL_COLOR_PROFILE_TARGET_ORIGINAL = 0, JXL_COLOR_PROFILE_TARGET_DATA = 1 } JxlColorProfileTarget;

/* jpegxl-sys/src/metadata/codestream_header.rs:286-388 (frame header of a displayed frame or non-coalesced layer) */
typedef enum { JXL_BLEND_REPLACE = 0, JXL_BLEND_ADD = 1, JXL_BLEND_BLEND = 2, JXL_BLEND_MULADD = 3, JXL_BLEND_MUL = 4 } JxlBlendMode;
typedef struct { JxlBlendMode blendmode; uint32_t source, alpha; JXL_BOOL clamp; } JxlBlendInfo;
typedef struct { JXL_BOOL have_crop; int32_t crop_x0, crop_y0; uint32_t xsize, ysize; JxlBlendInfo blend_info; uint32_t save_as_reference; } JxlLayerInfo;
typedef struct { uint32_t duration, timecode, name_length; JXL_BOOL is_last; JxlLayerInfo layer_info; } JxlFrameHeader;

typedef struct JxlDecoderStruct JxlDecoder;

/* jpegxl-sys/src/threads/parallel_runner.rs:46-122 */
typedef int JxlParallelRetCode;
typedef JxlParallelRetCode (*JxlParallelRunInit)(void* jpegxl_opaque, size_t num_threads);
typedef void (*JxlParallelRunFunction)(void* jpegxl_opaque, uint32_t value, size_t thread_id);
typedef JxlParallelRetCode (*JxlParallelRunner)(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init,
                                                JxlParallelRunFunction func, uint32_t start_range, uint32_t end_range);

/* ---- live decoder entry points (the 22 symbols jpegxl-rs/src/decode.rs calls) ----------------- */
uint32_t JxlDecoderVersion(void);                                                             /* decode.rs:370 -> 11002 */
JxlSignature JxlSignatureCheck(const uint8_t* buf, size_t len);                               /* decode.rs:385 */
JxlDecoder* JxlDecoderCreate(const JxlMemoryManager* memory_manager);                         /* decode.rs:400 */
void JxlDecoderReset(JxlDecoder* dec);                                                        /* decode.rs:408 */
void JxlDecoderDestroy(JxlDecoder* dec);                                                      /* decode.rs:414 */
JxlDecoderStatus JxlDecoderSetParallelRunner(JxlDecoder* dec, JxlParallelRunner runner, void* opaque);   /* decode.rs:487 */
JxlDecoderStatus JxlDecoderSubscribeEvents(JxlDecoder* dec, int events_wanted);               /* decode.rs:525 */
JxlDecoderStatus JxlDecoderSetKeepOrientation(JxlDecoder* dec, JXL_BOOL skip_reorientation);  /* decode.rs:563 */
JxlDecoderStatus JxlDecoderSetUnpremultiplyAlpha(JxlDecoder* dec, JXL_BOOL unpremul_alpha);   /* decode.rs:585 */
JxlDecoderStatus JxlDecoderSetRenderSpotcolors(JxlDecoder* dec, JXL_BOOL render_spotcolors);  /* decode.rs:602 */
/* decode.rs:622.  coalescing = FALSE: every regular frame of the image arrives as coded — JXL_DEC_FRAME, JXL_DEC_NEED_IMAGE_OUT_BUFFER (a buffer of
 * the FRAME's size: JxlDecoderImageOutBufferSize follows), JXL_DEC_FULL_IMAGE per frame — its pixels after the colour transform, not blended,
 * cropped frames at their own size; JxlDecoderGetFrameHeader says where it goes and how it blends. */
JxlDecoderStatus JxlDecoderSetCoalescing(JxlDecoder* dec, JXL_BOOL coalescing);
void JxlDecoderRewind(JxlDecoder* dec);                                                       /* decode.rs:438: back to the start, settings kept; set the input again */
void JxlDecoderSkipFrames(JxlDecoder* dec, size_t amount);                                    /* decode.rs:457: the next `amount` frames are not announced nor decoded */
JxlDecoderStatus JxlDecoderSkipCurrentFrame(JxlDecoder* dec);                                 /* decode.rs:472: after JXL_DEC_FRAME: do not decode this frame */
JxlDecoderStatus JxlDecoderGetFrameHeader(const JxlDecoder* dec, JxlFrameHeader* header);     /* decode.rs:1040: valid after JXL_DEC_FRAME */
JxlDecoderStatus JxlDecoderGetFrameName(const JxlDecoder* dec, char* name, size_t size);      /* decode.rs:1058 */
JxlDecoderStatus JxlDecoderGetExtraChannelBlendInfo(const JxlDecoder* dec, size_t index, JxlBlendInfo* blend_info);   /* decode.rs:1077 */
JxlDecoderStatus JxlDecoderProcessInput(JxlDecoder* dec);                                     /* decode.rs:662 — runs the HIP hot path */
JxlDecoderStatus JxlDecoderSetInput(JxlDecoder* dec, const uint8_t* data, size_t size);
/* jpegxl-sys/src/decode.rs:706: releases the input set by JxlDecoderSetInput; returns the number of bytes the decoder has not consumed
 * (everything while the headers could not be parsed yet, 0 afterwards: the decoder keeps its own copy of the codestream). */
size_t JxlDecoderReleaseInput(JxlDecoder* dec);       /* decode.rs:680 */
void JxlDecoderCloseInput(JxlDecoder* dec);                                                   /* decode.rs:724 */
JxlDecoderStatus JxlDecoderGetBasicInfo(const JxlDecoder* dec, JxlBasicInfo* info);           /* decode.rs:738 */
JxlDecoderStatus JxlDecoderGetICCProfileSize(const JxlDecoder* dec, JxlColorProfileTarget target, size_t* size);              /* decode.rs:862 */
JxlDecoderStatus JxlDecoderGetColorAsICCProfile(const JxlDecoder* dec, JxlColorProfileTarget target, uint8_t* icc, size_t size); /* decode.rs:884 */
JxlDecoderStatus JxlDecoderSetDesiredIntensityTarget(JxlDecoder* dec, float desired_intensity_target);                         /* decode.rs:921 */
JxlDecoderStatus JxlDecoderImageOutBufferSize(const JxlDecoder* dec, const JxlPixelFormat* format, size_t* size);              /* decode.rs:1100 */
JxlDecoderStatus JxlDecoderSetImageOutBuffer(JxlDecoder* dec, const JxlPixelFormat* format, void* buffer, size_t size);        /* decode.rs:1123 */
// jpegxl-sys/src/color/color_encoding.rs:125-159 and metadata/codestream_header.rs:247-279 (read-only descriptions of the image; jpegxl-rs itself reads JxlBasicInfo and the ICC profile only)
typedef struct {
  int color_space;                 // JxlColorSpace: 0 RGB, 1 grey, 2 XYB, 3 unknown
  int white_point;                 // JxlWhitePoint: 1 D65, 2 custom, 10 E, 11 DCI
  double white_point_xy[2];
  int primaries;                   // JxlPrimaries: 1 sRGB, 2 custom, 9 BT.2100, 11 P3
  double primaries_red_xy[2], primaries_green_xy[2], primaries_blue_xy[2];
  int transfer_function;           // JxlTransferFunction: 1 BT.709, 2 unknown, 8 linear, 13 sRGB, 16 PQ, 17 DCI, 18 HLG, 65535 gamma
  double gamma;
  int rendering_intent;            // 0 perceptual, 1 relative, 2 saturation, 3 absolute
} JxlColorEncoding;
typedef struct {
  int type;                        // JxlExtraChannelType: 0 alpha, 1 depth, 2 spot colour, 3 selection mask, 4 black, 5 CFA, 6 thermal, 15 unknown, 16 optional
  uint32_t bits_per_sample, exponent_bits_per_sample, dim_shift, name_length;
  JXL_BOOL alpha_premultiplied;
  float spot_color[4];
  uint32_t cfa_channel;
} JxlExtraChannelInfo;
// decode.rs:833: the colour encoding as enumerated values (error for images that carry an ICC profile instead: JxlDecoderGetColorAsICCProfile is the way then)
JxlDecoderStatus JxlDecoderGetColorAsEncodedProfile(const JxlDecoder* dec, JxlColorProfileTarget target, JxlColorEncoding* color_encoding);
// decode.rs:756 / :777
JxlDecoderStatus JxlDecoderGetExtraChannelInfo(const JxlDecoder* dec, size_t index, JxlExtraChannelInfo* info);
JxlDecoderStatus JxlDecoderGetExtraChannelName(const JxlDecoder* dec, size_t index, char* name, size_t size);
// decode.rs:509: bytes of input that make JxlDecoderGetBasicInfo likely to succeed; decode.rs:1495: the whole image is always decoded (ratio 1)
size_t JxlDecoderSizeHintBasicInfo(const JxlDecoder* dec);
size_t JxlDecoderGetIntendedDownsamplingRatio(const JxlDecoder* dec);
// jpegxl-sys/src/decode.rs:999-1025: the preview image (JXL_DEC_PREVIEW_IMAGE / JXL_DEC_NEED_PREVIEW_OUT_BUFFER; JxlBasicInfo.have_preview, .preview) — the preview frame is
// decoded on the GPU like an image of its own.  jpegxl-rs itself never subscribes to it (decode.rs:334-347).
JxlDecoderStatus JxlDecoderPreviewOutBufferSize(const JxlDecoder* dec, const JxlPixelFormat* format, size_t* size);
JxlDecoderStatus JxlDecoderSetPreviewOutBuffer(JxlDecoder* dec, const JxlPixelFormat* format, void* buffer, size_t size);
/* Pixel output through a callback instead of a buffer (decode.rs:289-309, :1172): called once per row (x = 0, num_pixels = xsize) from
 * the thread inside JxlDecoderProcessInput after the image has been decoded; the row memory is only valid during the call. */
typedef void (*JxlImageOutCallback)(void* opaque, size_t x, size_t y, size_t num_pixels, const void* pixels);
JxlDecoderStatus JxlDecoderSetImageOutCallback(JxlDecoder* dec, const JxlPixelFormat* format, JxlImageOutCallback callback, void* opaque);   /* decode.rs:1172 */
/* decode.rs:1200 (callback types :324-361): the rows are handed out after the image has been decoded on the device — init(init_opaque, 1 thread,
 * xsize pixels per call), run(..., thread 0, x = 0, y, xsize, row) for every row, destroy. */
typedef void* (*JxlImageOutInitCallback)(void* init_opaque, size_t num_threads, size_t num_pixels_per_thread);
typedef void (*JxlImageOutRunCallback)(void* run_opaque, size_t thread_id, size_t x, size_t y, size_t num_pixels, const void* pixels);
typedef void (*JxlImageOutDestroyCallback)(void* run_opaque);
JxlDecoderStatus JxlDecoderSetMultithreadedImageOutCallback(JxlDecoder* dec, const JxlPixelFormat* format, JxlImageOutInitCallback init_callback,
                                                            JxlImageOutRunCallback run_callback, JxlImageOutDestroyCallback destroy_callback, void* init_opaque);
/* decode.rs:1224 / :1258: a separate plane for an extra channel (one sample per pixel in `format`'s type, num_channels ignored), any of the image's extra channels: each
 * plane costs one more pass of the frame with that channel in the alpha slot of the interleaved output. */
JxlDecoderStatus JxlDecoderExtraChannelBufferSize(const JxlDecoder* dec, const JxlPixelFormat* format, size_t* size, uint32_t index);
JxlDecoderStatus JxlDecoderSetExtraChannelBuffer(JxlDecoder* dec, const JxlPixelFormat* format, void* buffer, size_t size, uint32_t index);
/* decode.rs:1326-1470: the boxes of the container (JXL_DEC_BOX once per box, signature and codestream boxes included; contents through a caller buffer with
 * JXL_DEC_BOX_NEED_MORE_OUTPUT when it is full; `brob` boxes decompressed on request when the system has libbrotlidec).  Boxes up to the first codestream
 * box are announced before JXL_DEC_BASIC_INFO, the others after the last frame. */
typedef struct { char type[4]; } JxlBoxType;                                                  /* types.rs:147 */
JxlDecoderStatus JxlDecoderSetBoxBuffer(JxlDecoder* dec, uint8_t* data, size_t size);
size_t JxlDecoderReleaseBoxBuffer(JxlDecoder* dec);
JxlDecoderStatus JxlDecoderSetDecompressBoxes(JxlDecoder* dec, JXL_BOOL decompress);
JxlDecoderStatus JxlDecoderGetBoxType(JxlDecoder* dec, JxlBoxType* type, JXL_BOOL decompressed);
JxlDecoderStatus JxlDecoderGetBoxSizeRaw(JxlDecoder* dec, uint64_t* size);
JxlDecoderStatus JxlDecoderGetBoxSizeContents(JxlDecoder* dec, uint64_t* size);
/* decode.rs:1482: which JXL_DEC_FRAME_PROGRESSION events are wanted (0 frames, 1 DC, 2 last passes, 3 passes are accepted, the finer ones rejected like libjxl
 * does); frames are decoded whole on the device, so no progression event is ever emitted.  decode.rs:1513: nothing partial exists to flush: JXL_DEC_ERROR
 * ("no flush was done"), as libjxl answers when no new image data is available. */
JxlDecoderStatus JxlDecoderSetProgressiveDetail(JxlDecoder* dec, int detail);
JxlDecoderStatus JxlDecoderFlushImage(JxlDecoder* dec);
/* decode.rs:1528 (types.rs:111-144): integer output is scaled to the full range of the buffer's type by default (JXL_BIT_DEPTH_FROM_PIXEL_FORMAT = 0); FROM_CODESTREAM (1) /
 * CUSTOM (2): samples in [0, 2^bits - 1] (the write stage's multiplier), bits <= the sample type's; float buffers keep their values.  Call after the image out buffer is set;
 * the setting lasts for that buffer. */
typedef struct { int type; uint32_t bits_per_sample, exponent_bits_per_sample; } JxlBitDepth;
JxlDecoderStatus JxlDecoderSetImageOutBitDepth(JxlDecoder* dec, const JxlBitDepth* bit_depth);
JxlDecoderStatus JxlDecoderSetJPEGBuffer(JxlDecoder* dec, uint8_t* data, size_t size);        /* decode.rs:1283 */
size_t JxlDecoderReleaseJPEGBuffer(JxlDecoder* dec);                                          /* decode.rs:1305 */

/* ---- libjxl_threads (threads/thread_parallel_runner.rs:44-65, resizable_parallel_runner.rs:42-67) ---------------- */
JxlParallelRetCode JxlThreadParallelRunner(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init, JxlParallelRunFunction func,
                                           uint32_t start_range, uint32_t end_range);
void* JxlThreadParallelRunnerCreate(const JxlMemoryManager* memory_manager, size_t num_worker_threads);
void JxlThreadParallelRunnerDestroy(void* runner_opaque);
size_t JxlThreadParallelRunnerDefaultNumWorkerThreads(void);
JxlParallelRetCode JxlResizableParallelRunner(void* runner_opaque, void* jpegxl_opaque, JxlParallelRunInit init, JxlParallelRunFunction func,
                                              uint32_t start_range, uint32_t end_range);
void* JxlResizableParallelRunnerCreate(const JxlMemoryManager* memory_manager);
void JxlResizableParallelRunnerSetThreads(void* runner_opaque, size_t num_threads);
uint32_t JxlResizableParallelRunnerSuggestThreads(uint64_t xsize, uint64_t ysize);
void JxlResizableParallelRunnerDestroy(void* runner_opaque);

/* The remaining symbols declared by jpegxl-sys (encoder, CMS, gain map, output colour profile ...) are exported as
 * error-returning stubs so the Rust crate links (csrc/jxl_stubs.cc, generated by tools/gen_stubs.py). */

/* ---- extension: device-resident batch decode ------------------------------------------------------------------- */
typedef struct JxlHipBatchStruct JxlHipBatch;
typedef struct {
  float lf_ms, lfpost_ms, hf_ms, idct_ms, filter_ms, out_ms, total_ms;
} JxlHipStageTimes;

/* Last error message of the calling thread ("" if none). */
const char* JxlHipLastError(void);
/* Lets `batch` use `owner`'s coefficient and pixel planes instead of allocating its own (they are only touched by part 2 of a
 * decode, so two batches whose part-2 halves run one after the other on one stream can share them: halves device memory of a
 * double-buffered pipeline).  Call before JxlHipBatchPrepare(batch); owner must be prepared, at least as large, and outlive batch. */
int JxlHipBatchShareBuffers(JxlHipBatch* batch, JxlHipBatch* owner);
/* The same for the quantised-coefficient planes (written by the HF stage of a decode, consumed and zeroed again by its IDCT stage): batches
 * whose [HF ... IDCT] intervals never overlap may use one set.  A deep pipeline alternates between two owners, so that the HF stage of
 * batch k + 1 runs beside the IDCT of batch k.  Call before JxlHipBatchPrepare(batch); owner must be prepared, at least as large, and
 * outlive batch.  Without it every batch holds planes of its own (106 MB per 4K frame). */
int JxlHipBatchShareCoefficients(JxlHipBatch* batch, JxlHipBatch* owner);
/* Host-only: parses the signature/container and image header of `data` and writes the ICC profile JxlDecoderGetColorAsICCProfile
 * would return (pass icc_out == NULL to query *icc_size).  Needs no GPU.  Returns 0 on success. */
int JxlHipColorProfileFromHeaders(const uint8_t* data, size_t size, uint8_t* icc_out, size_t* icc_size);
/* Host-only (needs no GPU): the dequantisation table (1 / weight, libjxl's coefficient layout) the decoder uses for library-default
 * quantisation kind `kind` (0..16, quant_weights.h), channel c (0 X, 1 Y, 2 B).  Writes min(n, cap) floats, returns n (0 on error). */
size_t JxlHipLibraryQuantTable(int kind, int c, float* out, size_t cap);
/* Host-only (needs no GPU): the serialisation half of JPEG reconstruction — parses a `jbrd` box payload and writes the JPEG file from
 * quantised coefficients (components x blocks x 64 int16, natural order, 4:4:4) and quantisation tables (components x 64, natural order).
 * *out_size: capacity in, bytes needed / written out.  Returns 0 on success, 2 if the buffer is too small, 1 on error. */
int JxlHipDebugWriteJpeg(const uint8_t* jbrd, size_t jbrd_size, uint32_t width, uint32_t height, const int16_t* coefficients, const int32_t* quant_tables,
                         uint8_t* out, size_t* out_size);
/* Same with JPEG sampling factors: sampling = (h, v) per component (NULL = all 1x1); the component planes follow one another, each
 * (mcu_rows * v) x (mcu_cols * h) blocks, the MCU grid being ceil(size / (8 * max factor)). */
int JxlHipDebugWriteJpegSampled(const uint8_t* jbrd, size_t jbrd_size, uint32_t width, uint32_t height, const uint32_t* sampling, const int16_t* coefficients,
                                const int32_t* quant_tables, uint8_t* out, size_t* out_size);
/* Host-only (no GPU): parse everything the host side parses and describe the image / its frames, one line each, into out
   (NUL-terminated, truncated to cap).  0 = accepted, 1 = rejected (JxlHipLastError()). */
int JxlHipDebugDescribe(const uint8_t* data, size_t size, char* out, size_t cap);
/* Creates a batch bound to HIP device `device`. */
JxlHipBatch* JxlHipBatchCreate(int device);
void JxlHipBatchDestroy(JxlHipBatch* batch);
/* Parses one image (headers, TOC, global tables) and appends it; returns its index or -1. */
int JxlHipBatchAddImage(JxlHipBatch* batch, const uint8_t* data, size_t size);
/* The same for `n` images at once, parsed on `num_threads` host threads (the per-image work — container, headers, TOC, entropy-code tables
 * of the global sections — is independent) and appended in order.  Returns the index of the first image, -1 on error (nothing is appended
 * then).  What a streaming caller feeds fresh compressed frames with, step after step. */
int JxlHipBatchAddImages(JxlHipBatch* batch, const uint8_t* const* datas, const size_t* sizes, int n, int num_threads);
/* Forgets the images of the batch but keeps its device arenas, its pinned staging buffer and the sharing set up with JxlHipBatchShare*:
 * the object can be filled (AddImage(s), SetOutput) and prepared again without allocating.  No decode of the old content may be in flight. */
void JxlHipBatchReset(JxlHipBatch* batch);
/* Basic info / required output size of image `index` for `format`. */
JxlDecoderStatus JxlHipBatchGetBasicInfo(const JxlHipBatch* batch, int index, JxlBasicInfo* info);
JxlDecoderStatus JxlHipBatchOutBufferSize(const JxlHipBatch* batch, int index, const JxlPixelFormat* format, size_t* size);
/* Output format (and optional caller-owned *device* destination; NULL = batch-owned) of image `index`. */
JxlDecoderStatus JxlHipBatchSetOutput(JxlHipBatch* batch, int index, const JxlPixelFormat* format, void* device_buffer);
/* Decode-thread packing: lanes between active entropy-decode threads (64 = one stream per wavefront, 1 = 64 per wavefront). */
void JxlHipBatchSetLaneStride(JxlHipBatch* batch, int lf, int hf);
/* Tuning / testing knobs: "force_generic_idct", "hf_block_threads", "lds_code_budget", "debug_stop_after", "lf_wide_once" (the next LF stage of the batch takes the
 * one-wavefront-per-stream kernel whatever the lane stride: shorter latency on an idle GPU), "lf_wp_narrow_test" (testing: the SIMT LF kernel's
 * weighted-predictor lanes hand a stream back to the one-wavefront-per-stream kernel at |sample| > 16 instead of 2^20). Unknown names are ignored. */
void JxlHipBatchSetOption(JxlHipBatch* batch, const char* name, int value);
/* Uploads streams and tables (inputs become HBM-resident) and allocates work buffers.  hip_stream: hipStream_t or NULL. */
JxlDecoderStatus JxlHipBatchPrepare(JxlHipBatch* batch, void* hip_stream);
/* Enqueues the decode of the whole batch on hip_stream (no host sync). */
JxlDecoderStatus JxlHipBatchDecode(JxlHipBatch* batch, void* hip_stream);
/* Same, bracketing every stage with HIP events recorded on hip_stream (still no host sync). */
JxlDecoderStatus JxlHipBatchDecodeTimed(JxlHipBatch* batch, void* hip_stream);
/* Enqueues one part of a decode: 0 = everything, 1 = front (LF decode, LF post-processing), 2 = rest (HF decode, IDCT,
 * filters, output); the rest may also be enqueued in two pieces, 3 = HF decode, 4 = IDCT, filters, output, and so may the front,
 * 5 = LF decode (all the HF decode needs), 6 = LF post-processing (needed by piece 4 only); piece 4 in turn as 7 = IDCT (the last stage
 * that touches the coefficient planes) and 8 = restoration filters, colour, write.  Front and rest may
 * go to different streams; the caller orders them with events, which lets the latency-bound LF stage of later batches overlap
 * the other stages of the current one (bench.py: three batches in flight, LF stages on two side streams). */
JxlDecoderStatus JxlHipBatchDecodePart(JxlHipBatch* batch, void* hip_stream, int part, int timed);
/* Waits for all timed decodes so far; returns per-stage sums in ms and the number of timed decodes. */
JxlDecoderStatus JxlHipBatchCollectTimes(JxlHipBatch* batch, JxlHipStageTimes* times, int* runs);
/* Waits for completion and checks per-frame device status. */
JxlDecoderStatus JxlHipBatchFinish(JxlHipBatch* batch, void* hip_stream);
/* Device pointer of the decoded pixels of image `index`; copy to host. */
void* JxlHipBatchDeviceOutput(const JxlHipBatch* batch, int index);
JxlDecoderStatus JxlHipBatchCopyOutput(JxlHipBatch* batch, int index, void* host_dst, size_t size, void* hip_stream);
/* Accounting for roofline reports. */
uint64_t JxlHipBatchTotalPixels(const JxlHipBatch* batch);
uint64_t JxlHipBatchCompressedBytes(const JxlHipBatch* batch);
/* Algorithmic (compulsory) HBM bytes of one decode per stage: lf, lfpost, hf, idct, filters, out. */
void JxlHipBatchStageBytes(const JxlHipBatch* batch, uint64_t out[6]);
uint64_t JxlHipBatchDeviceBytes(const JxlHipBatch* batch);
/* Facts about a prepared batch, by name (-1: unknown name): "lf_simt_frames" / "lf_legacy_frames" = VarDCT frames whose LF-group streams
 * take the SIMT kernel (one stream per lane, lane stride < 64) / the one-wavefront-per-stream kernel, "lf_simt_lanes", "lf_simt_waves", "lf_simt_wp" (1: the SIMT launch is the instantiation with weighted-predictor state), "hf_nonzeros" = non-zero AC coefficients of one decode of the batch (after a Finish). */
int64_t JxlHipBatchGetInfo(const JxlHipBatch* batch, const char* name);
/* Testing: after a decode, copies a device buffer of image `index`'s first (VarDCT) frame to the host — "plane_a" / "plane_b" (the padded
 * float XYB planes the stages ping-pong between, 8 bw x 8 bh samples per channel), "lf" / "llf" / "lfq", "inv_sigma", "blk_info", "coef_off" (one
 * value per 8x8 block), "coeff" (dense quantised coefficients, 65536 per group), "ytox" / "ytob".  With JxlHipBatchSetOption("debug_stop_after", s)
 * the tail of the decode stops after stage s (1 IDCT, 2 gaborish, 3 / 4 / 5 EPF pass 0 / 1 / 2; stage-by-stage filters: "force_unfused_filters"),
 * which is how tests/test_stage_formulas.py compares single stages with float64 restatements of their defining formulas.
 * Returns the size of the buffer in bytes (0 on error), copies min(size, cap). */
size_t JxlHipBatchDebugRead(JxlHipBatch* batch, int index, const char* name, int channel, void* dst, size_t cap, void* hip_stream);

/* ---- extension: the streaming decode pipeline of one GPU ----------------------------------------------------------------------
 * No counterpart in the reference: jpegxl-rs decodes batches by looping decode_with over files (jpegxl-rs/benches/decode.rs:16-19).  A pipeline object
 * owns what keeps a GPU busy with a stream of compressed images: a ring of batch objects, the coefficient sets and pixel planes all jobs share, its own
 * non-blocking HIP streams, the host threads that parse / build tables / upload / enqueue the latency-bound LF stage several jobs ahead, and the thread
 * that issues HF stages and tails in submission order (csrc/pipeline.h).  bench.py's headline is produced by these calls.
 * Concurrent callers of the libjxl API above share one such pipeline per device (csrc/scheduler.cc). */
typedef struct JxlHipPipelineStruct JxlHipPipeline;
typedef struct {
  int32_t jobs_in_flight;    /* jobs in flight on the GPU; 0 = default (11) */
  int32_t lf_streams;        /* side streams for the LF stages; 0 = default (11) */
  int32_t hf_streams;        /* HF stages in flight beside the tail, one coefficient set each + one; 0 = default (2) */
  int32_t prepare_threads;   /* host threads that each prepare one job at a time; 0 = default (3) */
  int32_t parse_threads;     /* host threads one job's images are parsed on; 0 = default (8) */
  int32_t lane_stride_lf, lane_stride_hf;   /* see JxlHipBatchSetLaneStride; 0 = defaults (8, 1) */
  int32_t wide_first;        /* LF stages at the start of a cold pipeline that take the one-wavefront-per-stream kernel; < 0 = default (4) */
  int32_t small_job_frames;  /* jobs of at most this many frames always take it, and sparse wavefronts in the HF stage (latency over occupancy); 0 = never */
  int32_t timed;             /* bracket the stages with HIP events: JxlHipPipelineCollectTimes */
  int32_t reserve_frames, reserve_width, reserve_height;   /* size the shared planes for jobs of this shape at creation (0: grown when the pipeline is idle) */
  int32_t reserve_plane_sets; /* 2: the frames take the stage-by-stage restoration filters (anything but gaborish + one EPF pass) and need a second set of pixel planes */
} JxlHipPipelineOptions;
JxlHipPipeline* JxlHipPipelineCreate(int device, const JxlHipPipelineOptions* options /* NULL = defaults */);
void JxlHipPipelineDestroy(JxlHipPipeline* pipeline);
/* Submits a job of n compressed images, all decoded to `format`.  Exactly one of device_out / host_out is given: n caller-owned destinations (device memory /
 * host memory — pinned, JxlHipHostAlloc, for full-speed copies that overlap later jobs), each at least JxlHipImageOutSize bytes; out_capacity (optional) is
 * checked per image.  The compressed bytes and the destinations must stay valid until JxlHipPipelineWait(ticket) returns.  Blocks while the ring is full.
 * An image that does not parse or decode fails alone.  Returns the job's ticket (>= 0) or -1 (JxlHipLastError). */
int64_t JxlHipPipelineSubmit(JxlHipPipeline* pipeline, const uint8_t* const* datas, const size_t* sizes, int n, const JxlPixelFormat* format, void* const* device_out,
                             void* const* host_out, const size_t* out_capacity);
/* Waits until the job has left the GPU (pixels written, host copies done).  image_status[i] (n entries, optional): 0 decoded, 1 failed; *end_ms (optional): when the
 * job's last byte was written, ms after JxlHipPipelineResetClock.  JXL_DEC_SUCCESS if every image decoded, else JXL_DEC_ERROR (JxlHipLastError names the first).
 * A ticket can be waited for once; jobs complete in submission order. */
JxlDecoderStatus JxlHipPipelineWait(JxlHipPipeline* pipeline, int64_t ticket, int* image_status, int n, float* end_ms);
JxlDecoderStatus JxlHipPipelineWaitAll(JxlHipPipeline* pipeline);        /* every job submitted so far has left the GPU (results stay collectable) */
JxlDecoderStatus JxlHipPipelineResetClock(JxlHipPipeline* pipeline);     /* waits for idle; end_ms and the prepare-time counters start over */
JxlDecoderStatus JxlHipPipelineCollectTimes(JxlHipPipeline* pipeline, JxlHipStageTimes* times, int* runs);   /* options.timed: per-stage sums (ms) over the jobs since the last call */
void JxlHipPipelineStageBytes(JxlHipPipeline* pipeline, uint64_t out[6]);   /* algorithmic bytes per stage of the job prepared last (JxlHipBatchStageBytes) */
/* "jobs", "slots", "coefficient_sets", "device_bytes", "shared_big_bytes", "shared_coef_bytes", "private_plane_jobs" (jobs that did not fit the shared planes),
 * "prepare_us_total" / "prepared_jobs" (host time of the prepare threads since the last clock reset), and of the job prepared / finished last: "frames",
 * "compressed_bytes", "total_pixels", "hf_nonzeros", "lf_simt_frames", "lf_legacy_frames", "lf_simt_wp".  -1: unknown name. */
int64_t JxlHipPipelineGetInfo(JxlHipPipeline* pipeline, const char* name);
/* Pinned host memory for host_out destinations (hipHostMalloc / hipHostFree). */
void* JxlHipHostAlloc(size_t bytes);
void JxlHipHostFree(void* p);
/* Host-only (needs no GPU): basic info and output size of an image for `format` from its headers — what a caller of JxlHipPipelineSubmit sizes its buffers with. */
JxlDecoderStatus JxlHipImageOutSize(const uint8_t* data, size_t size, const JxlPixelFormat* format, JxlBasicInfo* info, size_t* out_size);
/* Device arenas that batches and pipelines let go of are pooled per process (hipMalloc / hipFree of tens of GB cost seconds): JXL_HIP_ARENA_POOL_MB bounds the pool
 * (default 60 % of the device's memory, 0 = off); Trim hands every pooled block back to the runtime — for processes that share the GPU with another allocator —
 * and returns the bytes released; Held = bytes pooled right now. */
size_t JxlHipArenaPoolTrim(void);
size_t JxlHipArenaPoolHeld(void);
/* The scheduler behind the libjxl API (one shared pipeline per device; JXL_HIP_SCHEDULER=0 turns it off): jobs submitted / images decoded so far; Shutdown joins its
 * threads and frees its pipelines (tests). */
void JxlHipSchedulerStats(int device, int64_t* jobs, int64_t* images);
void JxlHipSchedulerShutdown(void);

/* ---- extension: the gather of decoded pixels over RCCL / xGMI (csrc/gather.cc) --------------------------------------------------
 * The multi-GPU path shards independent frames over one process per GPU and has ONE exchange step: the decoded pixels go to the consumer rank.  These
 * calls put it behind the C ABI so that a caller needs no PyTorch: rank 0 makes an id, the job hands it to the other ranks (file, environment, MPI ...), every rank
 * creates its communicator (world = 1 needs no id exchange and no RCCL).  librccl.so is loaded on first use.  All calls return 0 on success (JxlHipLastError). */
#define JXL_HIP_COMM_ID_BYTES 128
typedef struct JxlHipCommStruct JxlHipComm;
int JxlHipCommGetUniqueId(uint8_t id[JXL_HIP_COMM_ID_BYTES]);
JxlHipComm* JxlHipCommCreate(int device, int rank, int world, const uint8_t id[JXL_HIP_COMM_ID_BYTES]);
void JxlHipCommDestroy(JxlHipComm* comm);
/* Every rank holds `frames` decoded frames of frame_bytes each at `send` (device memory); `root` receives them at recv[rank][frame] (device memory, world x frames x
 * frame_bytes; ignored elsewhere) as point-to-point messages of chunk_frames frames, all peers of a chunk in one group — each over its own xGMI link — enqueued on
 * hip_stream.  The root's own shard is a device copy (skipped when send already points into recv). */
int JxlHipGatherFrames(JxlHipComm* comm, const void* send, size_t frame_bytes, int frames, void* recv, int root, int chunk_frames, void* hip_stream);
/* Shards of unequal length (1024 frames over 3, 5, 6, 7 GPUs): rank r holds frames_per_rank[r] frames, the root receives them in rank order (the frame order of the
 * unsharded job). */
int JxlHipGatherFramesRagged(JxlHipComm* comm, const void* send, size_t frame_bytes, const int* frames_per_rank, void* recv, int root, int chunk_frames, void* hip_stream);
/* Sum over the ranks, in place (per-rank consumers exchange checksums, not pixels). */
int JxlHipAllReduceSumI64(JxlHipComm* comm, int64_t* device_values, size_t count, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* JXL_HIP_H_ */
