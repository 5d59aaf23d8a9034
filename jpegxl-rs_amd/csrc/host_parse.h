// jxl-hip: host-side header parser.  Parses only the *structural* parts of a codestream — container, image and frame
// headers, TOC and the two global sections' tables (MA tree, entropy-code histograms → alias tables, quantisation
// parameters, coefficient orders).  Every O(pixels)/O(tokens) stream (LF coefficients, HF metadata, AC coefficients,
// modular channels) is left for the HIP kernels.  Replaces the header side of libjxl's decode.cc / headers.cc /
// frame_header.cc / toc.cc / dec_ans.cc(DecodeHistograms) / quant_weights.cc, reached by the reference through
// JxlDecoderProcessInput (jpegxl-rs/src/decode.rs:238) and JxlDecoderGetBasicInfo (decode.rs:246).
#pragma once
#include "jxl_dev.h"
#include "mm_alloc.h"
#include <string>
#include <vector>
#include <stdexcept>

namespace jxlhip {

struct ParseError : std::runtime_error {
  bool unsupported;
  ParseError(const std::string& s, bool unsup = false) : std::runtime_error(s), unsupported(unsup) {}
};

struct HostCode {  // entropy code in host memory, device-layout tables
  vec<uint8_t> ctx_map;
  vec<uint32_t> cfg;
  vec<uint64_t> alias;
  vec<uint16_t> pfx_count, pfx_syms;
  vec<uint32_t> pfx_sym_off;
  uint32_t num_ctx = 0, num_clusters = 0, log_alpha = 0, use_prefix = 0;   // num_ctx: without the LZ77 distance context
  bool lz77 = false;
  uint32_t lz_min_symbol = 0, lz_min_length = 0, lz_len_cfg = 0;
  DevCode View() const {
    DevCode d;
    d.ctx_map = ctx_map.data(); d.cfg = cfg.data(); d.alias = alias.data();
    d.pfx_count = pfx_count.data(); d.pfx_sym_off = pfx_sym_off.data(); d.pfx_syms = pfx_syms.data();
    d.num_ctx = num_ctx; d.num_clusters = num_clusters; d.log_alpha = log_alpha; d.use_prefix = use_prefix;
    d.lz77 = lz77; d.lz_min_symbol = lz_min_symbol; d.lz_min_length = lz_min_length; d.lz_len_cfg = lz_len_cfg;
    return d;
  }
};

struct HostTree {
  vec<TreeNode> nodes;
  uint32_t num_leaves = 0;
  bool uses_wp = false;
  int max_prop = 0;
};

struct BitDepthInfo { bool is_float = false; uint32_t bits = 8, exp_bits = 0; };
struct ExtraChannel { uint32_t type = 0; BitDepthInfo depth; uint32_t dim_shift = 0; bool alpha_associated = false; float spot[4] = {0, 0, 0, 0}; uint32_t cfa_channel = 1; vec<char> name; };   // spot: colour + solidity of a spot-colour channel (type 2)

struct ImageHeader {
  uint32_t xsize = 0, ysize = 0;
  uint32_t orientation = 1;
  uint32_t intrinsic_x = 0, intrinsic_y = 0;
  bool have_preview = false, have_animation = false, have_timecodes = false;
  uint32_t preview_x = 0, preview_y = 0;     // headers.cc PreviewHeader (the preview frame itself is skipped: nothing in jpegxl-rs asks for it)
  uint32_t tps_num = 0, tps_den = 0, num_loops = 0;
  BitDepthInfo depth;
  vec<ExtraChannel> extra;
  bool xyb_encoded = true;
  // colour encoding
  bool color_default = true, want_icc = false;
  uint32_t color_space = 0, white_point = 1, primaries = 1, tf = 13, rendering_intent = 1;
  bool have_gamma = false; uint32_t gamma = 0;
  double white_xy[2] = {0.3127, 0.3290}, prim_xy[6] = {0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204};  // custom CIE xy
  float intensity_target = 255.f, min_nits = 0.f, linear_below = 0.f;
  bool relative_to_max_display = false;
  float opsin_inv[9]; float opsin_bias[3]; float quant_bias[4];
  vec<float> up_weights[3];   // custom upsampling weights for 2x / 4x / 8x (15 / 55 / 210 values); empty = library default
  bool have_container = false;
  vec<uint8_t> icc;          // embedded ICC profile (want_icc), decoded
};

struct SqueezeStep { uint32_t horizontal, in_place, begin_c, num_c; };
struct TransformDesc {
  uint32_t id = 0, begin_c = 0, rct_type = 0, num_c = 0, nb_colors = 0, nb_deltas = 0, predictor = 0;
  vec<SqueezeStep> squeeze;
};

struct LoopFilterParams {
  uint32_t gab = 1, epf_iters = 2;
  float gab_w[6] = {0.115169525f, 0.061248592f, 0.115169525f, 0.061248592f, 0.115169525f, 0.061248592f};
  float sharp_lut[8] = {0.f, 1.f / 7, 2.f / 7, 3.f / 7, 4.f / 7, 5.f / 7, 6.f / 7, 1.f};
  float channel_scale[3] = {40.0f, 5.0f, 3.5f};
  float quant_mul = 0.46f, pass0_sigma_scale = 0.9f, pass2_sigma_scale = 6.5f, border_sad_mul = 2.0f / 3.0f;
};

struct QuantTableSpec {
  uint32_t mode = 0;
  uint32_t num_bands = 0; float bands[3][17];
  uint32_t num_bands4 = 0; float bands4[3][17];
  float idw[3][3], dct2w[3][6], dct4mul[3][2], dct4x8mul[3], afvw[3][9];
  float raw_den = 0; vec<int32_t> raw[3];
};

struct Section { uint64_t offset, size; };

// ---- image features and frame compositing (LfGlobal / frame header), host-parsed: dec_patch_dictionary.cc, splines.cc, dec_noise.cc
struct BlendInfoH { uint32_t mode = 0, alpha_channel = 0, source = 0; bool clamp = false; };   // BlendMode: 0 replace, 1 add, 2 blend, 3 mul-add, 4 mul
struct PatchBlendH { uint32_t mode = 0, alpha_channel = 0, clamp = 0; };                        // PatchBlendMode 0..7
struct PatchPosH { int64_t x = 0, y = 0; vec<PatchBlendH> blend; };                      // blend[0] colour, blend[1 + e] extra channel e
struct PatchRefH { uint32_t ref = 0, x0 = 0, y0 = 0, xsize = 0, ysize = 0; vec<PatchPosH> pos; };
struct SplineH { vec<std::pair<int64_t, int64_t>> control_points; int32_t color_dct[3][32]; int32_t sigma_dct[32]; };
struct FrameFeatures {
  vec<PatchRefH> patches;
  int32_t spline_quant_adjust = 0;
  vec<SplineH> splines;
  vec<std::pair<int64_t, int64_t>> spline_start;
  bool has_noise = false;
  float noise_lut[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};
// One rendered spline sample (splines.cc SplineSegment) and the per-row draw lists built on the host (host_features.cc)
struct SplineSegmentDev { float center_x, center_y, maximum_distance, inv_sigma, sigma_over_4_times_intensity, color[3]; };
struct SplineDrawList { vec<SplineSegmentDev> segments; vec<uint32_t> row_start; vec<uint32_t> indices; };
void BuildSplineDrawList(const FrameFeatures& f, float y_to_x, float y_to_b, uint32_t height, SplineDrawList* out);
float FastPowf(float base, float exponent);   // base/fast_math-inl.h (host_parse.cc)

struct FramePlan {
  // frame header
  uint32_t frame_type = 0; bool modular = false; uint64_t flags = 0; bool do_ycbcr = false;
  uint32_t upsampling = 1, group_size_shift = 1, x_qm_scale = 3, b_qm_scale = 2;
  uint32_t num_passes = 1; uint32_t pass_shift[11] = {0};
  // passes.h GetDownsamplingBracket: the Modular channels PassGroup (pass, g) carries are those with pass_min_shift <= min(hshift, vshift) <= pass_max_shift
  uint32_t num_ds = 0, downsample[4] = {0}, ds_last_pass[4] = {0};
  int32_t pass_min_shift[11] = {0}, pass_max_shift[11] = {2};
  uint32_t mod_pass = 0;                   // VarDCT frames: the pass whose sections carry the (unsqueezed) extra channels
  uint32_t ModUnitPasses() const { return modular ? num_passes : 1; }      // Modular sub-streams per group the device decodes
  uint32_t NumModUnits() const { return num_lf_groups + num_groups * ModUnitPasses(); }
  bool is_last = true;
  // LF frames (frame_header.cc kDCFrame / kUseDcFrame): a frame of type 1 is the LF image — one sample per 8x8 block — of the frames one level below it
  // (lf_level 1: of the regular frames); a frame with use_lf_frame has no LF coefficients of its own and reads the LF frame of level lf_level + 1
  uint32_t lf_level = 0; bool use_lf_frame = false;
  LoopFilterParams lf;
  // compositing: position / size of the frame on the image canvas, blending, reference slots (frame_header.cc)
  bool have_crop = false; int32_t x0 = 0, y0 = 0;
  uint32_t frame_w = 0, frame_h = 0;       // frame size after upsampling (width / height below are the coded size)
  BlendInfoH blend; vec<BlendInfoH> ec_blend;
  uint32_t duration = 0, timecode = 0, save_as_reference = 0; bool save_before_ct = false;
  vec<char> name;                          // frame name (UTF-8, no terminator)
  float sigma_for_modular = 1.0f;
  FrameFeatures feat;
  uint64_t frame_end_bitpos = 0;           // first bit after the frame's last section (= next frame header)
  bool partial = false;                    // the codestream ends inside the frame's PassGroup sections: LF part complete; the AC groups that are completely there are decoded (progressive flush)
  uint32_t partial_ac_sections = 0;        // (partial) PassGroup sections that are completely there
  // Modular frames: every unit's group header has been looked at (ParseLocalModularStreams) — whether any unit carries transforms of its own (then it decodes into per-unit
  // scratch) or a local tree with the weighted predictor: what decoder.cc sizes the per-unit scratch by (an 8192x8192 frame has 1040 units)
  bool mod_units_scanned = false, mod_local_transforms = false, mod_local_wp = false;
  // geometry
  uint32_t width = 0, height = 0, group_dim = 256;
  uint32_t xgroups = 0, ygroups = 0, num_groups = 0, xlfgroups = 0, ylfgroups = 0, num_lf_groups = 0;
  uint32_t bw = 0, bh = 0;  // 8x8 blocks
  // chroma subsampling of YCbCr frames (frame_header.h YCbCrChromaSubsampling): sampling-factor mode per channel (Cb, Y, Cr) and the
  // shifts relative to the largest factor; bw / bh are padded to whole cells of the coarsest channel
  uint32_t jpeg_upsampling[3] = {0, 0, 0}, hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
  bool subsampled = false;
  // TOC
  vec<Section> sections;   // byte offsets into the codestream buffer
  bool single_section = false;
  // LfGlobal
  float m_lf[3] = {1.0f / 4096, 1.0f / 512, 1.0f / 256};
  uint32_t global_scale = 1, quant_lf = 1;
  BlockCtxDev bcm;
  uint32_t color_factor = 84; float base_x = 0.f, base_b = 1.f; int32_t ytox_lf = 0, ytob_lf = 0;
  bool has_global_tree = false;
  HostTree tree;
  HostCode tree_code;       // code of the global-tree streams
  // GlobalModular: channel list after global transforms; channel data starts at global_data_bitpos
  struct ModChannel { uint32_t w, h; int32_t hshift, vshift; };
  vec<ModChannel> gchannels;
  uint32_t nb_meta_channels = 0;
  vec<TransformDesc> gtransforms;
  WPHeader gwp;
  bool g_use_global_tree = true;
  // Modular sub-streams with an MA tree and an entropy code of their own (GroupHeader.use_global_tree = 0): unit 0 = the global
  // stream, 1 + u = LfGroup / PassGroup unit u (LF groups first).  Parsed on the host for Modular frames, whose sections start
  // with their stream; data_bitpos = absolute bit position of the stream's ANS state.
  struct LocalStream { uint32_t unit = 0; HostTree tree; HostCode code; uint64_t data_bitpos = 0; };
  vec<LocalStream> local_streams;
  int max_prop = 0;                  // largest property index any MA tree of the frame tests
  uint64_t global_data_bitpos = 0;   // absolute bit position (codestream) where the global stream's ANS state starts
  uint32_t global_decodable = 0;     // number of leading channels decoded in the global section
  uint32_t nb_color_channels = 0;    // colour channels held in the modular image (0 for VarDCT)
  // HfGlobal
  QuantTableSpec qspec[17];
  uint32_t num_hf_presets = 1;
  vec<uint32_t> used_orders;                // per pass
  vec<vec<uint16_t>> custom_order;  // [pass*39 + bucket*3 + c] (empty = natural)
  vec<HostCode> ac_code;                    // per pass
  uint64_t end_bitpos = 0;                          // where parsing of the last host-parsed section stopped
};

// Aligned, padded copy of the codestream (container stripped) — also the H2D staging buffer
struct Codestream {
  vec<uint32_t> storage;
  size_t size = 0;
  Codestream() = default;
  Codestream(const Codestream&) = default;
  Codestream(Codestream&&) = default;
  Codestream& operator=(const Codestream&) = default;
  Codestream& operator=(Codestream&&) = default;
  // large copies (>= 1 MB, plain malloc) go back to a process-wide free list instead of the C library: a 73 MB codestream is mapped, page-faulted in and unmapped again
  // per decode otherwise — with three prepare workers x eight parse threads faulting at once, 100-190 ms per job of eight such frames (host_parse.cc TakeCodestreamStorage)
  ~Codestream();
  const uint8_t* data() const { return reinterpret_cast<const uint8_t*>(storage.data()); }
  uint8_t* data() { return reinterpret_cast<uint8_t*>(storage.data()); }
  size_t padded_size() const { return storage.size() * 4; }
};

enum SigResult { kSigNotEnoughBytes = 0, kSigInvalid = 1, kSigCodestream = 2, kSigContainer = 3 };
SigResult CheckSignature(const uint8_t* buf, size_t len);

// Extracts the codestream; returns false if more input is needed (truncated container).
// container boxes JPEG reconstruction draws on: `jbrd`, the first `Exif` and the first `xml ` box; *_brob: the payload is still the Brotli
// stream of a `brob` box (decompressed where it is used, jpeg_recon.cc)
struct MetadataBoxes { vec<uint8_t> jbrd, exif, xml; bool have_exif = false, have_xml = false, exif_brob = false, xml_brob = false; };
bool ExtractCodestream(const uint8_t* data, size_t size, Codestream* cs, bool* have_container, bool* has_jbrd, MetadataBoxes* boxes = nullptr);

// Parses the image header; *frame_bitpos receives the bit position of the first frame header.
void ParseImageHeader(const Codestream& cs, ImageHeader* ih, uint64_t* frame_bitpos);

// Parses frame header + TOC + LfGlobal (tables only).  On return plan->sections is filled and, for multi-section
// frames, HfGlobal has been parsed too.  For single-section frames HfGlobal must be parsed later with ParseHfGlobal
// at the bit position where the device-decoded LfGroup streams end.
// header_and_toc_only: the frame is only stepped over (the preview frame): frame header and TOC give plan->frame_end_bitpos, nothing else is looked at
void ParseFrameStart(const Codestream& cs, const ImageHeader& ih, uint64_t frame_bitpos, FramePlan* plan, bool header_and_toc_only = false, bool allow_partial = false);
void ParseHfGlobal(const Codestream& cs, const ImageHeader& ih, uint64_t bitpos, FramePlan* plan);

// Dequantisation table (1/weight) of quant kind `kind`, channel c; natural coefficient order of a strategy.
void ComputeQuantTable(const QuantTableSpec& spec, int kind, int c, vec<float>* out);
vec<uint16_t> NaturalCoeffOrder(int strategy);
// ICC v4.4 matrix/TRC profile of the enumerated colour encoding (icc_profile.cc); throws ParseError.
vec<uint8_t> SynthesizeIcc(const ImageHeader& ih);
// CIE xy of the image's white point (2 values) and primaries (6: red, green, blue), enumerated or custom (JxlColorEncoding, jpegxl-sys color_encoding.rs:125-159)
void ColorChromaticities(const ImageHeader& ih, double white_xy[2], double primaries_xy[6]);
std::string ColorDescription(const ImageHeader& ih);
// dec_xyb.cc OutputEncodingInfo::SetColorEncoding: linear sRGB -> the image's own primaries / white point (row-major 3x3, through XYZ D50
// with linear Bradford adaptation) and the luminance weights of those primaries.  Returns false — identity, sRGB luminances — for
// sRGB / D65, grey images and images that carry an ICC profile.
bool SrgbToOriginalPrimaries(const ImageHeader& ih, double m[9], float luminances[3]);
extern const uint8_t kBucketStrategy[13];
extern const uint8_t kKindRows[17], kKindCols[17];

}  // namespace jxlhip
