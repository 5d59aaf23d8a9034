// jxl-hip: bit-exact JPEG reconstruction (jpegxl-rs/src/decode.rs:493-514 `reconstruct`, libjxl lib/jxl/jpeg/{dec_jpeg_data.cc,
// jpeg_data.cc, dec_jpeg_data_writer.cc}).  A JPEG that was transcoded losslessly carries, in the container's `jbrd` box, everything
// of the original file that is not coefficient data (marker order, APPn / COM payloads, Huffman tables, scan script, restart and
// padding details); the quantised coefficients come out of the VarDCT entropy decode on the GPU, the Huffman re-encode and the
// marker serialisation below run on the host.
#pragma once
#include <cstdint>
#include <string>
#include <utility>
#include <vector>
#include "mm_alloc.h"

namespace jxlhip {

struct JpegHuffmanCode { uint32_t slot_id = 0; bool is_last = true; uint32_t counts[17] = {0}; vec<uint32_t> values; };
struct JpegScanComponent { uint32_t comp_idx = 0, ac_tbl_idx = 0, dc_tbl_idx = 0; };
struct JpegScanInfo {
  uint32_t num_components = 1, Ss = 0, Se = 63, Al = 0, Ah = 0, last_needed_pass = 0;
  JpegScanComponent components[4];
  vec<uint32_t> reset_points;
  vec<std::pair<uint32_t, uint32_t>> extra_zero_runs;   // (block index, number of extra 0xF0 symbols)
};
struct JpegQuantTable { uint32_t precision = 0, index = 0; bool is_last = true; int32_t values[64] = {0}; };   // values: natural (row-major) order
struct JpegComponentInfo { uint32_t id = 0, quant_idx = 0, h_samp = 1, v_samp = 1; };
struct JpegData {
  vec<uint8_t> marker_order;
  vec<vec<uint8_t>> app_data; vec<uint32_t> app_marker_type;
  vec<vec<uint8_t>> com_data, inter_marker_data;
  vec<uint8_t> tail_data;
  vec<JpegQuantTable> quant;
  vec<JpegComponentInfo> components;
  vec<JpegHuffmanCode> huffman_code;
  vec<JpegScanInfo> scan_info;
  uint32_t restart_interval = 0;
  bool has_zero_padding_bit = false;
  vec<uint8_t> padding_bits;
};

// Parses the payload of a `jbrd` box (bit-packed JPEGData bundle, then one Brotli stream with the marker payloads; Brotli comes
// from the system's libbrotlidec.so.1, loaded at run time).  Returns false with *err set when the data is malformed or Brotli is
// unavailable.  APPn markers of type ICC / Exif / XMP come back sized, with marker byte, length and tag in place (dec_jpeg_data.cc) and
// their payload still to be filled by FillJpegMetadata.
bool ParseJbrd(const uint8_t* data, size_t size, JpegData* jd, std::string* err);

// The payloads jbrd leaves out because the JPEG XL file holds them elsewhere: the ICC profile of the codestream, spread over the APP2
// chunks in order (decode.cc SetJPEGDataFromICC), the `Exif` box minus its 4-byte TIFF offset and the `xml ` box (decode_to_jpeg.cc
// SetExif / SetXmp); sizes must match what the markers announce.  *_brob: the box arrived Brotli-compressed (`brob`).
struct JpegMetadataSources {
  const uint8_t* icc = nullptr; size_t icc_size = 0;
  const uint8_t* exif = nullptr; size_t exif_size = 0; bool exif_brob = false;
  const uint8_t* xml = nullptr; size_t xml_size = 0; bool xml_brob = false;
};
bool FillJpegMetadata(JpegData* jd, const JpegMetadataSources& src, std::string* err);

// Serialises the JPEG: markers in jbrd order, quantisation tables as filled in by the caller (jd.quant[i].values), entropy-coded
// scans from the quantised coefficients.  coeffs[c]: (mcu_rows * v_samp) x (mcu_cols * h_samp) x 64 int16 in natural order for component
// c, the MCU grid being ceil(size / (8 * max sampling factor)); sampling factors from jd.components (set by the caller from the frame header).
// Sequential (baseline / extended) and progressive (spectral selection, successive approximation, EOB runs) Huffman scans.
bool WriteJpeg(const JpegData& jd, uint32_t width, uint32_t height, const int16_t* const* coeffs, vec<uint8_t>* out, std::string* err);

// One-shot Brotli decompression of a stream whose plain size is not announced (container `brob` boxes): the output buffer grows until the
// stream fits or `limit` bytes are exceeded.  False when the system has no libbrotlidec or the stream is damaged.
bool BrotliDecompressAll(const uint8_t* data, size_t size, size_t limit, vec<uint8_t>* out);

}  // namespace jxlhip
