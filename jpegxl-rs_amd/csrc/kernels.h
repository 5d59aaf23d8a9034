// jxl-hip: device-side frame descriptor + kernel launch interface (implemented in kernels.hip).
#pragma once
#include "jxl_dev.h"
#if !defined(__HIPCC__)
// host-only builds of the library's host sources (the sanitizer build, tests/test_sanitizers.py): the runtime API and the vector types, no device code
#include <hip/hip_runtime_api.h>
#include <hip/hip_vector_types.h>
#endif

namespace jxlhip {

// One channel of a Modular frame's global image as the sub-streams see it (after the global transforms' MetaApply):
// plane location in the work arena, dimensions and the squeeze shifts that decide which section carries it.
struct ModChanDev { uint64_t off; uint32_t w, h; int32_t hshift, vshift; };

// Per-pass tables of a VarDCT frame (HfPass in the codestream): entropy code, coefficient orders, left shift of the values.
struct PassDev { DevCode code; const uint16_t* orders[39]; uint32_t shift; uint32_t pad; };

// One per frame of a batch; array lives in device memory.  All pointers are device pointers.
// A Modular sub-stream that carries its own MA tree and entropy code (GroupHeader.use_global_tree = 0, parsed on the host for
// Modular frames, where every section starts with its stream).
struct ModLocalDev { const TreeNode* tree; uint32_t tree_nodes, uses_wp, max_prop, pad; uint64_t data_bitpos; DevCode code; };

struct FrameDev {
  // geometry
  uint32_t width, height, bw, bh, xgroups, ygroups, num_groups, xlfgroups, num_lf_groups, cw, ch;
  uint32_t group_dim, is_modular, plane_stride, plane_rows;
  // codestream + sections (byte offsets); for single-section frames the *_bitpos fields are used instead
  const uint8_t* cs;
  uint64_t cs_size;
  const uint64_t* sec_off;
  const uint64_t* sec_size;
  uint32_t single_section;
  uint64_t lf_start_bitpos;      // single-section: where LfGroup starts
  uint64_t hf_start_bitpos;      // single-section: where PassGroup starts
  uint64_t* stream_end_bitpos;   // [0] = end of LfGroup stream (single-section), [1] = end of global modular stream
  // entropy codes / tree
  const TreeNode* tree;
  uint32_t tree_nodes;
  DevCode mod_code;
  uint32_t uses_wp;
  WPHeader gwp;
  // chroma subsampling (YCbCr JPEG transcodes): channel c lives on a grid of (bw >> hs[c]) x (bh >> vs[c]) blocks, packed into the
  // top-left corner of the frame-sized LF / pixel planes; coefficients keep the offsets of the full-resolution block a cell starts at
  uint32_t hs[3], vs[3], subsampled;
  uint32_t tree_max_prop;         // largest property index in any MA tree of the frame (>= 16: previous-channel properties)
  const ModLocalDev* mod_local;   // Modular sub-streams with a tree and code of their own, indexed 0 = global stream, 1 + unit = LfGroup /
                                  // PassGroup unit (entries with tree == nullptr use the frame's tree); nullptr: none in this frame
  DevCode ac_code;
  const uint16_t* orders[39];
  const BlockCtxDev* bcm;
  uint32_t num_hf_presets, preset_bits;
  uint32_t num_passes;            // progressive passes: PassGroup section (p, g) adds value << passes[p].shift to the coefficients
  const PassDev* passes;          // [num_passes]; ac_code / orders above are pass 0's
  // dequant
  float lf_fac[3], cfl_lf_x, cfl_lf_b;
  float inv_global_scale, x_dm, b_dm, quant_bias[4], color_scale, base_x, base_b;
  const float* qtable[17 * 3];
  uint32_t skip_lf_smoothing;
  // loop filter + colour
  uint32_t gab, epf_iters;
  float gab_w[9];                 // per channel: n0, n1, n2 (normalised)
  float epf_sharp_lut[8], epf_channel_scale[3], epf_quant_mul, epf_quant_scale;
  float epf_sm[3], epf_bsm[3];    // per pass sad multipliers (normal, border)
  float opsin_inv[9], neg_bias[3], neg_bias_cbrt[3];
  uint32_t color_mode;            // 0: XYB->sRGB, 1: XYB->linear, 2: YCbCr->RGB, 3: none (RGB as is), 4: XYB->gamma (FastPowf), 5: XYB->Rec.709, 6: XYB->PQ, 7: XYB->HLG
  uint32_t is_gray;
  // VarDCT buffers
  int32_t* lfq[3];
  float* lf[3];
  float* lf_tmp[3];
  float* llf[3];
  uint32_t* blk_info;
  uint32_t* coef_off;
  uint2* vb_list;                 // per group: up to 1024 {strategy | log2 cx << 5 | log2 cx cy << 8 | order bucket << 12 | x << 16 | y << 21 | block-context bucket << 26, coefficient offset}
  uint32_t* vb_count;             // per group: number of varblocks
  int8_t* ytox; int8_t* ytob;
  int32_t* coeff[3];
  float* plane_a[3];
  float* plane_b[3];
  float* inv_sigma;
  int32_t* lf_scratch;            // per LF group scratch (HF metadata channels)
  uint64_t lf_scratch_stride;     // ints per LF group
  int32_t* wp_scratch;            // per stream WP state
  uint64_t wp_scratch_stride;
  // modular buffers
  const ModChanDev* mod_chan;     // channel table of the global image *before* undoing global transforms (meta channels first)
  uint8_t* mod_base;              // work arena base the table's offsets refer to
  uint32_t mod_nchan;
  uint32_t mod_nb_meta;
  uint32_t mod_global_decodable;
  uint64_t mod_global_bitpos;
  int32_t* mod_group_scratch;     // per group scratch
  uint64_t mod_group_scratch_stride;
  uint32_t mod_bits;
  int32_t* mod_wp_scratch;        // weighted-predictor state of the Modular sub-streams: global, then one per LfGroup / PassGroup unit
  uint64_t mod_wp_stride;
  uint64_t* hf_end_bitpos;        // VarDCT frames with extra channels: where each group's HF coefficient stream ended (its Modular part starts there)
  const int32_t* alpha_plane;     // VarDCT frames: decoded alpha extra channel (image-sized), or null
  float alpha_factor;             // 1 / (2^bits - 1)
  // output
  uint8_t* out;
  uint64_t out_stride;            // bytes per row
  uint32_t out_channels, out_type /*0 u8 1 u16 2 f32 3 f16*/, out_big_endian;
  float out_int_mul;              // integer output: sample = round(clamp(v, 0, 1) x this) — 255 / 65535, or 2^bits - 1 (JxlDecoderSetImageOutBitDepth)
  uint32_t out_orient;            // 1..8: orientation applied while writing (1 = none)
  // upsampling (frame coded at 1/upsampling of the image size): width/height above are the CODED size
  uint32_t upsampling, img_w, img_h;   // img_*: image size the write stage covers (= width/height when upsampling == 1)
  const float* up_weights;        // 15 / 55 / 210 coefficients of the symmetric (5N x 5N) kernel matrix, N = upsampling / 2
  float* up_plane[4];             // upsampled X, Y, B (and alpha as float) planes, img_w x img_h
  uint32_t mod_cfg_uniform;       // the hybrid-integer configuration shared by every cluster of mod_code, or 0xFFFFFFFF
  uint4* place_rec;               // varblock placement records (one per varblock, grouped per band: BandRecordBase), bw x bh entries
  uint32_t* place_cnt;            // [LF group * 8 + band] records of the band
  uint32_t* band_start;           // [LF group * 8 + band] index of the band's first entry in the LF group's strategy list (0xFFFFFFFF: damaged)
  uint32_t mod_unit_passes;       // Modular sub-streams per group: the frame's passes (Modular frames), 1 (VarDCT frames: the extra channels' pass only, mod_pass)
  uint32_t mod_pass;
  int32_t pass_min_shift[11], pass_max_shift[11];   // passes.h GetDownsamplingBracket per pass: the channels (by min(hshift, vshift)) a PassGroup of that pass carries
  uint32_t lz_lf_base;            // LZ77-coded LF-group streams of a VarDCT frame: index of LF group 0's window in lz_window (the Modular units' windows come first)
  uint32_t use_lf_frame;          // frame_header.cc kUseDcFrame: no LF coefficients in the LfGroups, the LF image is an LF frame's samples (no dequantisation, no smoothing, LF context 0)
  uint32_t lf_simt;               // the LF-group streams of this frame are decoded by LfDecodeSimtKernel (one stream per lane), placement by LfPlaceKernel
  uint32_t* status;
  uint32_t* frame_flags;          // [0] != 0: some varblock is not contained in a 64x64 tile (generic IDCT path)
  uint32_t* hf_written;           // running count of non-zero AC coefficients the HF stage wrote for this frame (bench accounting)
  uint32_t* lz_ac_window;         // LZ77-coded AC streams: kAcLzWindow entries per group stream (reused pass after pass); null otherwise
  uint32_t* lz_window;            // LZ77-coded Modular streams: 2^20-entry windows, one per stream (global, then LfGroup / PassGroup units)
  uint32_t post_mode;             // 1: the frame ends in its float planes (after the restoration filters); upsampling, colour transform and the
                                  // write stage are done by the host-planned frame tail (kernels_features.hip) — multi-frame images, image features
  float inverse_gamma;            // colour modes 4 / 5 (XYB -> gamma / Rec.709 transfer)
  float hdr_par[5];               // colour mode 6 (PQ): [0] intensity_target / 10000; 7 (HLG): [0] OOTF exponent, [1] apply it, [2..4] luminances
};

struct LaunchCfg {
  int lane_stride_lf = 64;   // lanes between active decode threads (64 = one stream per wave, 1 = one per lane)
  int lane_stride_hf = 64;
  int lane_stride_mod = 64;
  int any_wp = 0;            // some MA tree of the batch uses the weighted predictor (the Modular kernels then reserve LDS for its state)
  int any_subsampled = 0;    // some frame is chroma-subsampled (its own IDCT kernel; the SIMT HF kernel only)
  int any_multipass = 0;     // some frame has progressive passes (the general instantiation of the SIMT HF kernel)
  int any_prefix_ac = 0;     // some VarDCT frame's AC code is a prefix code or LZ77-coded (HfDecodeKernel beside the SIMT kernel)
  int any_local_trees = 0;   // some Modular sub-stream carries its own MA tree / code (second launch of the group kernel)
  // filled by Batch::Prepare: LDS needs of the batch (bytes of cfg + ctx map + alias tables, MA-tree nodes)
  int max_tree_nodes = 1024, mod_code_bytes = 1 << 20, ac_code_bytes = 1 << 20, ac_code_bytes_compact = 1 << 20;   // (compact: 6 bytes per alias slot, StageCodeCompact)
  int force_generic_idct = 0;
  // filled by Batch::Finish from the per-frame flags the LF stage sets (deterministic per stream): once known, the
  // kernels for varblocks outside a 64x64 tile / the DCT128-256 family are only launched when some frame needs them
  int lf_wide_once = 0;              // the next LF launch takes the one-wavefront-per-stream kernel for every frame, SIMT-eligible or not (a pipeline that starts
                                     // on an idle GPU: 100 instead of 250 ms until the first batch can go on); reset by the launch
  int lf_wp_narrow_test = 0;         // testing: the SIMT LF kernel's weighted-predictor lanes hand a stream back at |sample| > 16 instead of 2^20
  int lf_force_big = 0;              // testing: the launch shape of LfDecodeKernel (kernels.hip LaunchLfDecode)
  int lf_head_start = 0;             // the LF launch waits until the next HF launch is resident (set for pipelined front-only calls)
  int idct_flags_known = 0, any_irregular_blocks = 1, any_big_blocks = 1;
  int hf_lanes_per_wave = 0, hf_lanes_per_wg = 0;   // SIMT HF decode: group streams per wavefront / per workgroup (0: the throughput defaults — a frame's streams on four wavefronts of one workgroup)
  int skip_hf = 0;                   // the HF stage is left out: every AC coefficient stays zero (progressive flush at the kDC step, decoder.h AddImage allow_partial)
  int max_passes = 0;                // > 0: at most this many passes of every group are decoded (progressive flush at a kLastPasses / kPasses step)
  int no_flag_wait = 0;              // latency mode (pipeline.h small-job scheduler): the tail never waits on the host for the placement flags of the LF stage, every IDCT kernel variant is launched
  int need_tile4_plain = 1, need_tile4_special = 1, need_tile8_plain = 1, need_tile8_special = 1;   // IdctTileKernel<TB, SPECIAL> variants some frame takes
  int debug_stop_after = 0;          // testing: the tail of a decode stops after 1 = IDCT, 2 = gaborish, 3 / 4 / 5 = EPF pass 0 / 1 / 2 (stage-by-stage filters only)
  int force_unfused_filters = 0;     // testing: stage-by-stage gaborish / EPF / output kernels even for fusable frames        // testing: run the generic (non-tiled) IDCT kernel even for tile-regular frames
  int hf_block_threads = 512;        // threads per HF-decode block (streams per block = threads / lane_stride_hf)
  int lds_code_budget = 64 * 1024;   // bytes of LDS the entropy-code tables (cfg, ctx map, alias) may take per block
};

void InitDeviceTables(void* stream);

// ---- SIMT LF decode (LfDecodeSimtKernel): one LF-group stream (three LF coefficient channels + four HF-metadata channels, ~275 000 tokens
// for a 2048x2048 region) per LANE instead of per wavefront.  The host classifies every channel of every stream from the frame's MA tree
// (decoder.cc ClassifyLfChannel): after the static splits (channel index, stream id) and per class of rows (splits on the row, property 2)
// the subtree may test at most two of {W + N - NW (9), W (7), N (6), largest weighted-predictor error (15)} and its leaves share one
// predictor with offset 0, multiplier 1.  A row class is an 8-byte entry {table offset, class word}; its table — 1024 bytes either way —
// maps the clamped property value(s) to the cluster and is read through the L2 like the alias tables: nothing about a stream lives in LDS
// (measured in round 4: even 2 KB of LDS per LF workgroup — resident for hundreds of milliseconds on every CU — fragments the LDS the 80 KB
// HF workgroups need: HF stage 56 -> 71 ms, with the tables in LDS too 80 ms, although the LF stage itself got 12 % shorter).
// Class word: bits 0-1 kind (0: one cluster, bits 16-23; 1: table over property A, value in [-512, 511]; 2: table over A x B, values in
// [-16, 15]); bits 2-4 predictor (0 zero, 1 W, 2 N, 3 clamped gradient, 4 weighted, 5 (W + N) / 2, 6 select, 7 NE); bits 5-6 / 7-8
// property A / B (0: W + N - NW, 1: W, 2: N, 3: weighted-predictor error); bit 10: the channel keeps weighted-predictor state.
// Channel entry: a row class when every row of the channel has the same one; with bit 9 set, lut_off points at
// [512 bytes: row (clamped to 511) -> index][8-byte row classes].
constexpr uint32_t kLfSimtRows = 1u << 9, kLfSimtWpLive = 1u << 10;
constexpr uint32_t kLfSimtWpInts = 2 * 258 * 8;            // ints of weighted-predictor state per stream: 2 rows x 258 records {four sub-predictor errors, true error, pad}
constexpr int32_t kLfRedoMark = 0x5eed;                    // lf_scratch[2] of an LF group: the SIMT kernel handed the stream back (LfDecodeKernel decodes it again)
struct LfSimtChan { uint32_t lut_off; uint32_t info; };    // lut_off: byte offset in the batch's table blob; info: class word
struct LfSimtStream { uint32_t frame, group; LfSimtChan chan[7]; };   // channels: LF Y, X, B, then ytox, ytob, block info, sharpness
struct LfSimtLane { uint32_t first, count; };               // a lane decodes streams [first, first + count) one after the other
struct LfSimtPlan {
  const LfSimtStream* streams = nullptr; const LfSimtLane* lanes = nullptr; const uint8_t* luts = nullptr;   // device pointers
  const uint2* units = nullptr; uint32_t num_units = 0;     // varblock placement: {frame, LF group | band << 16} of every 32-row band of every VarDCT frame, longest first
  uint32_t num_lanes = 0, lanes_per_wave = 16;
  int any_legacy = 1;          // some VarDCT frame of the batch still takes LfDecodeKernel (one wavefront per stream)
  int any_general = 0;         // some SIMT stream needs more than the lean instantiation offers (kernels.hip LfDecodeSimtKernel<WP, GEN>)
  int any_wp = 0;              // some SIMT stream keeps weighted-predictor state (the kernel's WP instantiation; LfDecodeKernel follows for the streams it hands back)
};

// VarDCT stages.  max_* are maxima over the batch (grid sizing); nframes = frames in batch.
void LaunchLfDecode(const FrameDev* frames, int nframes, int max_lf_groups, const LaunchCfg& cfg, void* stream, const LfSimtPlan* simt = nullptr);
void LaunchLfPost(const FrameDev* frames, int nframes, int max_bw, int max_bh, int max_groups, void* stream);
void LaunchHfDecode(const FrameDev* frames, int nframes, int max_groups, const LaunchCfg& cfg, void* stream);
void LaunchZeroFailedCoefficients(const FrameDev* frames, int nframes, void* stream);   // behind the HF stage: a failed frame's coefficient planes go back to zero
void LaunchIdct(const FrameDev* frames, int nframes, int max_groups, int max_bw, int max_bh, const LaunchCfg& cfg, void* stream);
struct FilterPlan {
  bool any_upsampled = false;    // some frame needs UpsampleKernel before the write stage
  int max_out_w = 0, max_out_h = 0; bool any_fused = false, any_unfused = false, any_gab = false; int max_epf = 0; };   // over the VarDCT frames of a batch
void LaunchFilters(const FrameDev* frames, int nframes, int max_w, int max_h, const FilterPlan& fp, const LaunchCfg& cfg, void* stream);
void LaunchOutput(const FrameDev* frames, int nframes, int max_w, int max_h, const FilterPlan& fp, const LaunchCfg& cfg, void* stream);
// Modular stages
struct ModOutputArgs { const int32_t* color[3]; const int32_t* alpha; uint32_t ncolor; float color_factor, alpha_factor; uint32_t float_bits, float_exp_bits; };   // float_bits != 0: colour samples are float bit patterns
void LaunchModularGlobal(const FrameDev* frames, int nframes, const LaunchCfg& cfg, void* stream);
uint32_t ModularGroupLdsBytes(const LaunchCfg& cfg);
void LaunchModularGroups(const FrameDev* frames, int nframes, int max_units, const LaunchCfg& cfg, void* stream);   // max_units: most LF groups + groups x passes of a frame
// inverse Squeeze of one channel: (avg, res) -> out; horizontal: avg aw x h, res rw x h, out (aw+rw) x h; vertical: avg w x ah, res w x rh
void LaunchModInvSqueeze(const int32_t* avg, const int32_t* res, int32_t* out, int horizontal, uint32_t aw, uint32_t ah, uint32_t rw, uint32_t rh, void* stream);
// the same step of n <= kSqueezeBatch equally shaped, independent channels (one per image of a batch) in one launch
constexpr int kSqueezeBatch = 16;
struct SqueezeBatch { const int32_t* avg[kSqueezeBatch]; const int32_t* res[kSqueezeBatch]; int32_t* out[kSqueezeBatch]; };
void LaunchModInvSqueezeBatch(const SqueezeBatch& b, int n, int horizontal, uint32_t aw, uint32_t ah, uint32_t rw, uint32_t rh, void* stream);
void LaunchModRct(int32_t* a, int32_t* b, int32_t* c, size_t n, uint32_t rct_type, void* stream);
void LaunchModPalette(const int32_t* pal, int32_t* const* out, uint32_t nb_colors, uint32_t num_c, uint32_t bit_depth, size_t n, void* stream);
void LaunchChromaUpsample(const float* src, uint32_t src_stride, float* dst, uint32_t dst_stride, uint32_t cw, uint32_t ch, uint32_t hs, uint32_t vs, uint32_t out_w, uint32_t out_h, void* stream);
void LaunchModPaletteDelta(const int32_t* pal, int32_t* const* out, uint32_t nb_colors, uint32_t num_c, uint32_t bit_depth, uint32_t nb_deltas, uint32_t predictor,
                           uint32_t w, uint32_t h, const WPHeader& wp, int32_t* wp_scratch, uint32_t wp_stride, void* stream);
void LaunchModOutput(const FrameDev* frames, int fidx, const ModOutputArgs& a, int w, int h, void* stream);

// ---- frame tail of images with several frames or image features (kernels_features.hip); explicit arguments, device pointers ----
struct SplineSegmentDev;   // host_parse.h
// One placement of a patch: source rectangle (pointers already at its top-left sample) -> frame position (x, y).
// mode[k] = PatchBlendMode | alpha channel << 8 | clamp << 16 for k = 0 (colour), 1 + e (extra channel e)
struct PatchEntryDev { const float* src[3]; const float* esrc[4]; uint32_t src_stride, esrc_stride; int32_t x, y; uint32_t xs, ys; uint32_t mode[5]; uint32_t pad; };
struct PatchFrameArgs { float* p[3]; float* ec[4]; uint32_t stride, ec_stride, w, h, num_extra, premul_mask; };
struct NoiseArgs { float* p[3]; uint32_t stride, w, h; float* noise[3]; uint32_t noise_stride, group_dim, visible_frame_index, nonvisible_frame_index; float lut[8]; float ytox, ytob; };
// mode: 3 = transfer function only (after a spot-colour stage in linear light); 0 XYB -> linear -> transfer function (tf_kind 0 sRGB, 1 linear, 2 gamma, 3 Rec.709, 4 PQ, 5 HLG), 1 YCbCr -> RGB, 2 copy
struct ColorArgs { const float* src[3]; float* dst[3]; uint32_t src_stride, dst_stride, w, h, mode, tf_kind; float inverse_gamma, opsin_inv[9], neg_bias[3], neg_bias_cbrt[3], hdr_par[5]; };
// mode[k] = BlendMode | alpha channel << 8 | clamp << 16; bg pointers are null when the source slot is empty (treated as zeros)
struct BlendArgs {
  const float* fg[3]; const float* fg_ec[4]; uint32_t fg_stride, fg_ec_stride, fw, fh; int32_t x0, y0;
  const float* bg[3]; uint32_t bg_stride; const float* bg_alpha; uint32_t bg_alpha_stride;
  const float* bg_ec[4]; const float* bg_ec_alpha[4]; uint32_t bg_ec_stride[4];
  float* canvas[3]; float* canvas_ec[4]; uint32_t canvas_stride, canvas_ec_stride, img_w, img_h, num_extra, premul_mask; uint32_t mode[5];
};
struct WriteArgs { const float* p[3]; const float* alpha; uint32_t stride, alpha_stride, img_w, img_h; uint8_t* out; uint64_t out_stride; uint32_t out_channels, out_type, out_big_endian, out_orient, is_gray, unpremul; float out_int_mul; };
void LaunchIntToFloat(const int32_t* src, uint32_t src_stride, float* dst, uint32_t dst_stride, uint32_t w, uint32_t h, float factor, void* stream, uint32_t float_bits = 0,
                      uint32_t float_exp_bits = 0);   // float_bits != 0: the integers are float bit patterns (IntToFloatSample)
void LaunchXybModToFloat(const int32_t* cy, const int32_t* cx, const int32_t* cb, uint32_t src_stride, float* const dst[3], uint32_t dst_stride, uint32_t w, uint32_t h, const float fac[3], void* stream);
void LaunchPatches(const PatchFrameArgs& a, const PatchEntryDev* entries, const uint32_t* tile_start, const uint32_t* tile_list, void* stream);
void LaunchSplines(float* const p[3], uint32_t stride, uint32_t w, uint32_t h, const SplineSegmentDev* segs, const uint32_t* row_start, const uint32_t* indices, void* stream);
void LaunchUpsamplePlane(const float* src, uint32_t src_stride, uint32_t w, uint32_t h, float* dst, uint32_t dst_stride, uint32_t ow, uint32_t oh, uint32_t up, const float* weights, void* stream);
void LaunchNoise(const NoiseArgs& a, void* stream);
void LaunchColor(const ColorArgs& a, void* stream);
// stage_spot.cc: p[c] = mix * color[c] + (1 - mix) * p[c], mix = scale * spot
struct SpotArgs { float* p[3]; const float* spot; uint32_t stride, spot_stride, w, h; float color[3], scale; };
void LaunchSpot(const SpotArgs& a, void* stream);
void LaunchBlend(const BlendArgs& a, void* stream);
void LaunchWrite(const WriteArgs& a, void* stream);
// JPEG reconstruction: quantised coefficients of a JPEG-transcoded frame in JPEG layout — component c (0 Y, 1 Cb, 2 Cr), block raster
// order, 64 coefficients in natural (row-major) order: DC from the quantised LF image, AC from the coefficient planes with the integer
// chroma-from-luma of the transcoder undone (dec_group.cc, jpeg branch).  qt: the JPEG quantisation tables, natural order.
struct JpegCoefArgs { int16_t* out; uint32_t ncomp; int32_t qt[3][64]; uint32_t comp_off[3]; };   // comp_off: first block of a component's plane (subsampled frames)
void LaunchJpegCoefficients(const FrameDev* frames, int fidx, const JpegCoefArgs& a, uint32_t bw, uint32_t bh, void* stream);
void LaunchCopyPlane(const float* src, uint32_t src_stride, float* dst, uint32_t dst_stride, uint32_t w, uint32_t h, void* stream);

// names of the kernels (for profiling summaries)
extern const char* const kKernelNames[];

}  // namespace jxlhip
