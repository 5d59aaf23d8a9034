"""jpegxl-rs_amd — host-side mirror of the jpegxl-rs decoder API on top of the MI355X-native C ABI (include/jxl_hip.h).

The reference's host language (Rust) is not available in the build image, so the layer above the C ABI is mirrored
here in Python with the same names, argument meaning and error behaviour as jpegxl-rs:

    decoder_builder()                      -> jpegxl-rs/src/decode.rs:535  (bon builder; options are keyword args)
    JxlDecoder.decode(data)                -> decode.rs:440   (pixel type inferred from the header)
    JxlDecoder.decode_with(data, dtype)    -> decode.rs:461   (decode_with::<T>)
    JxlDecoder.reconstruct(data)           -> decode.rs:493
    Metadata / Pixels / PixelFormat        -> decode/result.rs:26-76, decode.rs:49-82
    DecodeError subclasses                 -> errors.rs:27-52
    ThreadsRunner / ResizableRunner        -> parallel/threads_runner.rs, parallel/resizable_runner.rs
    check_valid_signature                  -> utils.rs:25-33

Everything goes through ctypes into lib/libjxl.so (hand-written HIP kernels); there is no CPU fallback: without the
built library or without a GPU the calls fail loudly.
"""
import ctypes as C
import os
import sys
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_HERE, "lib")
LIBJXL_PATH = os.environ.get("JXL_HIP_LIBJXL") or os.path.join(LIB_DIR, "libjxl.so")      # (override: A/B runs of kernel variants)
LIBTHREADS_PATH = os.path.join(LIB_DIR, "libjxl_threads.so")

# ---- C types (include/jxl_hip.h) ------------------------------------------------------------------------------------
JXL_TYPE_FLOAT, JXL_TYPE_UINT8, JXL_TYPE_UINT16, JXL_TYPE_FLOAT16 = 0, 2, 3, 5
JXL_NATIVE_ENDIAN, JXL_LITTLE_ENDIAN, JXL_BIG_ENDIAN = 0, 1, 2
(JXL_DEC_SUCCESS, JXL_DEC_ERROR, JXL_DEC_NEED_MORE_INPUT, JXL_DEC_NEED_PREVIEW_OUT_BUFFER, JXL_DEC_NEED_IMAGE_OUT_BUFFER,
 JXL_DEC_JPEG_NEED_MORE_OUTPUT, JXL_DEC_BOX_NEED_MORE_OUTPUT) = 0, 1, 2, 3, 5, 6, 7
JXL_DEC_BASIC_INFO, JXL_DEC_COLOR_ENCODING, JXL_DEC_PREVIEW_IMAGE, JXL_DEC_FRAME = 0x40, 0x100, 0x200, 0x400
JXL_DEC_FULL_IMAGE, JXL_DEC_JPEG_RECONSTRUCTION, JXL_DEC_BOX, JXL_DEC_FRAME_PROGRESSION, JXL_DEC_BOX_COMPLETE = 0x1000, 0x2000, 0x4000, 0x8000, 0x10000


class JxlPixelFormat(C.Structure):
    _fields_ = [("num_channels", C.c_uint32), ("data_type", C.c_int), ("endianness", C.c_int), ("align", C.c_size_t)]


class JxlBasicInfo(C.Structure):
    _fields_ = [("have_container", C.c_int), ("xsize", C.c_uint32), ("ysize", C.c_uint32), ("bits_per_sample", C.c_uint32),
                ("exponent_bits_per_sample", C.c_uint32), ("intensity_target", C.c_float), ("min_nits", C.c_float),
                ("relative_to_max_display", C.c_int), ("linear_below", C.c_float), ("uses_original_profile", C.c_int),
                ("have_preview", C.c_int), ("have_animation", C.c_int), ("orientation", C.c_int32), ("num_color_channels", C.c_uint32),
                ("num_extra_channels", C.c_uint32), ("alpha_bits", C.c_uint32), ("alpha_exponent_bits", C.c_uint32),
                ("alpha_premultiplied", C.c_int), ("preview_xsize", C.c_uint32), ("preview_ysize", C.c_uint32),
                ("tps_numerator", C.c_uint32), ("tps_denominator", C.c_uint32), ("num_loops", C.c_uint32), ("have_timecodes", C.c_int),
                ("intrinsic_xsize", C.c_uint32), ("intrinsic_ysize", C.c_uint32), ("padding", C.c_uint8 * 100)]


class JxlMemoryManager(C.Structure):
    _fields_ = [("opaque", C.c_void_p), ("alloc", C.c_void_p), ("free", C.c_void_p)]


class JxlColorEncoding(C.Structure):
    """jpegxl-sys color/color_encoding.rs:125-159"""
    _fields_ = [("color_space", C.c_int), ("white_point", C.c_int), ("white_point_xy", C.c_double * 2), ("primaries", C.c_int), ("primaries_red_xy", C.c_double * 2),
                ("primaries_green_xy", C.c_double * 2), ("primaries_blue_xy", C.c_double * 2), ("transfer_function", C.c_int), ("gamma", C.c_double), ("rendering_intent", C.c_int)]


class JxlExtraChannelInfo(C.Structure):
    """jpegxl-sys metadata/codestream_header.rs:247-279"""
    _fields_ = [("type", C.c_int), ("bits_per_sample", C.c_uint32), ("exponent_bits_per_sample", C.c_uint32), ("dim_shift", C.c_uint32), ("name_length", C.c_uint32),
                ("alpha_premultiplied", C.c_int), ("spot_color", C.c_float * 4), ("cfa_channel", C.c_uint32)]


class JxlBlendInfo(C.Structure):
    """jpegxl-sys/src/metadata/codestream_header.rs:305-315"""
    _fields_ = [("blendmode", C.c_int), ("source", C.c_uint32), ("alpha", C.c_uint32), ("clamp", C.c_int)]


class JxlLayerInfo(C.Structure):
    """codestream_header.rs:323-353"""
    _fields_ = [("have_crop", C.c_int), ("crop_x0", C.c_int32), ("crop_y0", C.c_int32), ("xsize", C.c_uint32), ("ysize", C.c_uint32),
                ("blend_info", JxlBlendInfo), ("save_as_reference", C.c_uint32)]


class JxlFrameHeader(C.Structure):
    """codestream_header.rs:358-388"""
    _fields_ = [("duration", C.c_uint32), ("timecode", C.c_uint32), ("name_length", C.c_uint32), ("is_last", C.c_int), ("layer_info", JxlLayerInfo)]


class JxlHipStageTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("lf_ms", "lfpost_ms", "hf_ms", "idct_ms", "filter_ms", "out_ms", "total_ms")]


class JxlHipPipelineOptions(C.Structure):
    """include/jxl_hip.h JxlHipPipelineOptions (0 / negative = the library's default)"""
    _fields_ = [(n, C.c_int32) for n in ("jobs_in_flight", "lf_streams", "hf_streams", "prepare_threads", "parse_threads", "lane_stride_lf", "lane_stride_hf", "wide_first",
                                         "small_job_frames", "timed", "reserve_frames", "reserve_width", "reserve_height", "reserve_plane_sets")]


assert C.sizeof(JxlBasicInfo) == 204 and C.sizeof(JxlPixelFormat) == 24 and C.sizeof(JxlMemoryManager) == 24

_lib = None
_threads = None


class JxlBitDepth(C.Structure):
    """jpegxl-sys/src/common/types.rs:133-144"""
    _fields_ = [("type", C.c_int), ("bits_per_sample", C.c_uint32), ("exponent_bits_per_sample", C.c_uint32)]


class NativeLibraryMissing(RuntimeError):
    pass


def libjxl():
    """Loads lib/libjxl.so (the HIP decode path).  Fails loudly if it has not been built — there is no fallback."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIBJXL_PATH):
            raise NativeLibraryMissing(f"{LIBJXL_PATH} not built: run __graft_entry__.build() (hipcc --offload-arch=gfx950)")
        L = C.CDLL(LIBJXL_PATH, mode=C.RTLD_GLOBAL)
        vp, u8p, sz = C.c_void_p, C.c_char_p, C.c_size_t
        sig = {
            "JxlDecoderVersion": (C.c_uint32, []), "JxlSignatureCheck": (C.c_int, [u8p, sz]),
            "JxlDecoderCreate": (vp, [vp]), "JxlDecoderReset": (None, [vp]), "JxlDecoderDestroy": (None, [vp]),
            "JxlDecoderSetParallelRunner": (C.c_int, [vp, vp, vp]), "JxlDecoderSubscribeEvents": (C.c_int, [vp, C.c_int]),
            "JxlDecoderSetKeepOrientation": (C.c_int, [vp, C.c_int]), "JxlDecoderSetUnpremultiplyAlpha": (C.c_int, [vp, C.c_int]),
            "JxlDecoderSetRenderSpotcolors": (C.c_int, [vp, C.c_int]), "JxlDecoderSetCoalescing": (C.c_int, [vp, C.c_int]),
            "JxlDecoderProcessInput": (C.c_int, [vp]), "JxlDecoderSetInput": (C.c_int, [vp, vp, sz]), "JxlDecoderCloseInput": (None, [vp]),
            "JxlDecoderGetBasicInfo": (C.c_int, [vp, C.POINTER(JxlBasicInfo)]),
            "JxlDecoderGetICCProfileSize": (C.c_int, [vp, C.c_int, C.POINTER(sz)]),
            "JxlDecoderGetColorAsICCProfile": (C.c_int, [vp, C.c_int, vp, sz]),
            "JxlDecoderSetDesiredIntensityTarget": (C.c_int, [vp, C.c_float]),
            "JxlDecoderImageOutBufferSize": (C.c_int, [vp, C.POINTER(JxlPixelFormat), C.POINTER(sz)]),
            "JxlDecoderSetImageOutBuffer": (C.c_int, [vp, C.POINTER(JxlPixelFormat), vp, sz]),
            "JxlDecoderGetColorAsEncodedProfile": (C.c_int, [vp, C.c_int, C.POINTER(JxlColorEncoding)]),
            "JxlDecoderGetExtraChannelInfo": (C.c_int, [vp, sz, C.POINTER(JxlExtraChannelInfo)]),
            "JxlDecoderGetExtraChannelName": (C.c_int, [vp, sz, C.c_char_p, sz]),
            "JxlDecoderSizeHintBasicInfo": (sz, [vp]),
            "JxlDecoderGetIntendedDownsamplingRatio": (sz, [vp]),
            "JxlDecoderPreviewOutBufferSize": (C.c_int, [vp, C.POINTER(JxlPixelFormat), C.POINTER(sz)]),
            "JxlDecoderSetPreviewOutBuffer": (C.c_int, [vp, C.POINTER(JxlPixelFormat), vp, sz]),
            "JxlDecoderReleaseInput": (sz, [vp]), "JxlDecoderSetJPEGBuffer": (C.c_int, [vp, vp, sz]), "JxlDecoderReleaseJPEGBuffer": (sz, [vp]),
            "JxlDecoderGetFrameHeader": (C.c_int, [vp, C.POINTER(JxlFrameHeader)]), "JxlDecoderGetFrameName": (C.c_int, [vp, C.c_char_p, sz]),
            "JxlDecoderGetExtraChannelBlendInfo": (C.c_int, [vp, sz, C.POINTER(JxlBlendInfo)]), "JxlDecoderSkipFrames": (None, [vp, sz]),
            "JxlDecoderSkipCurrentFrame": (C.c_int, [vp]), "JxlDecoderRewind": (None, [vp]),
            "JxlDecoderSetMultithreadedImageOutCallback": (C.c_int, [vp, C.POINTER(JxlPixelFormat), vp, vp, vp, vp]),
            "JxlDecoderExtraChannelBufferSize": (C.c_int, [vp, C.POINTER(JxlPixelFormat), C.POINTER(sz), C.c_uint32]),
            "JxlDecoderSetExtraChannelBuffer": (C.c_int, [vp, C.POINTER(JxlPixelFormat), vp, sz, C.c_uint32]),
            "JxlDecoderSetBoxBuffer": (C.c_int, [vp, vp, sz]), "JxlDecoderReleaseBoxBuffer": (sz, [vp]),
            "JxlDecoderSetDecompressBoxes": (C.c_int, [vp, C.c_int]), "JxlDecoderGetBoxType": (C.c_int, [vp, C.c_char_p, C.c_int]),
            "JxlDecoderGetBoxSizeRaw": (C.c_int, [vp, C.POINTER(C.c_uint64)]), "JxlDecoderGetBoxSizeContents": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
            "JxlDecoderSetProgressiveDetail": (C.c_int, [vp, C.c_int]), "JxlDecoderFlushImage": (C.c_int, [vp]),
            "JxlDecoderSetImageOutBitDepth": (C.c_int, [vp, C.POINTER(JxlBitDepth)]),
            "JxlHipLastError": (C.c_char_p, []), "JxlHipBatchCreate": (vp, [C.c_int]), "JxlHipBatchDestroy": (None, [vp]),
            "JxlHipBatchAddImage": (C.c_int, [vp, vp, sz]), "JxlHipBatchGetBasicInfo": (C.c_int, [vp, C.c_int, C.POINTER(JxlBasicInfo)]),
            "JxlHipBatchOutBufferSize": (C.c_int, [vp, C.c_int, C.POINTER(JxlPixelFormat), C.POINTER(sz)]),
            "JxlHipBatchSetOutput": (C.c_int, [vp, C.c_int, C.POINTER(JxlPixelFormat), vp]),
            "JxlHipBatchSetLaneStride": (None, [vp, C.c_int, C.c_int]), "JxlHipBatchSetOption": (None, [vp, C.c_char_p, C.c_int]),
            "JxlHipBatchPrepare": (C.c_int, [vp, vp]), "JxlHipBatchDecode": (C.c_int, [vp, vp]),
            "JxlHipBatchDecodeTimed": (C.c_int, [vp, vp]), "JxlHipBatchFinish": (C.c_int, [vp, vp]),
            "JxlHipBatchDecodePart": (C.c_int, [vp, vp, C.c_int, C.c_int]),
            "JxlHipBatchCollectTimes": (C.c_int, [vp, C.POINTER(JxlHipStageTimes), C.POINTER(C.c_int)]),
            "JxlHipBatchDeviceOutput": (vp, [vp, C.c_int]), "JxlHipBatchCopyOutput": (C.c_int, [vp, C.c_int, vp, sz, vp]),
            "JxlHipBatchTotalPixels": (C.c_uint64, [vp]), "JxlHipBatchCompressedBytes": (C.c_uint64, [vp]),
            "JxlHipBatchStageBytes": (None, [vp, C.POINTER(C.c_uint64 * 6)]), "JxlHipBatchDeviceBytes": (C.c_uint64, [vp]),
            "JxlHipBatchGetInfo": (C.c_int64, [vp, C.c_char_p]), "JxlHipBatchReset": (None, [vp]),
            "JxlHipBatchDebugRead": (sz, [vp, C.c_int, C.c_char_p, C.c_int, vp, sz, vp]),
            "JxlHipBatchAddImages": (C.c_int, [vp, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_int, C.c_int]),
            "JxlHipBatchShareBuffers": (C.c_int, [vp, vp]), "JxlHipBatchShareCoefficients": (C.c_int, [vp, vp]),
            "JxlHipPipelineCreate": (vp, [C.c_int, C.POINTER(JxlHipPipelineOptions)]), "JxlHipPipelineDestroy": (None, [vp]),
            "JxlHipPipelineSubmit": (C.c_int64, [vp, C.POINTER(C.c_char_p), C.POINTER(sz), C.c_int, C.POINTER(JxlPixelFormat), C.POINTER(vp), C.POINTER(vp), C.POINTER(sz)]),
            "JxlHipPipelineWait": (C.c_int, [vp, C.c_int64, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_float)]),
            "JxlHipPipelineWaitAll": (C.c_int, [vp]), "JxlHipPipelineResetClock": (C.c_int, [vp]),
            "JxlHipPipelineCollectTimes": (C.c_int, [vp, C.POINTER(JxlHipStageTimes), C.POINTER(C.c_int)]),
            "JxlHipPipelineStageBytes": (None, [vp, C.POINTER(C.c_uint64 * 6)]), "JxlHipPipelineGetInfo": (C.c_int64, [vp, C.c_char_p]),
            "JxlHipHostAlloc": (vp, [sz]), "JxlHipHostFree": (None, [vp]),
            "JxlHipImageOutSize": (C.c_int, [vp, sz, C.POINTER(JxlPixelFormat), C.POINTER(JxlBasicInfo), C.POINTER(sz)]),
            "JxlHipArenaPoolTrim": (sz, []), "JxlHipArenaPoolHeld": (sz, []),
            "JxlHipCommGetUniqueId": (C.c_int, [vp]), "JxlHipCommCreate": (vp, [C.c_int, C.c_int, C.c_int, vp]), "JxlHipCommDestroy": (None, [vp]),
            "JxlHipGatherFrames": (C.c_int, [vp, vp, sz, C.c_int, vp, C.c_int, C.c_int, vp]),
            "JxlHipGatherFramesRagged": (C.c_int, [vp, vp, sz, C.POINTER(C.c_int), vp, C.c_int, C.c_int, vp]),
            "JxlHipAllReduceSumI64": (C.c_int, [vp, vp, sz, vp]),
            "JxlHipSchedulerStats": (None, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]), "JxlHipSchedulerShutdown": (None, []),
        }
        for name, (res, args) in sig.items():
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def libjxl_threads():
    global _threads
    if _threads is None:
        if not os.path.exists(LIBTHREADS_PATH):
            raise NativeLibraryMissing(f"{LIBTHREADS_PATH} not built: run __graft_entry__.build()")
        T = C.CDLL(LIBTHREADS_PATH)
        vp = C.c_void_p
        T.JxlThreadParallelRunnerCreate.restype = vp; T.JxlThreadParallelRunnerCreate.argtypes = [vp, C.c_size_t]
        T.JxlThreadParallelRunnerDestroy.argtypes = [vp]
        T.JxlThreadParallelRunnerDefaultNumWorkerThreads.restype = C.c_size_t
        T.JxlResizableParallelRunnerCreate.restype = vp; T.JxlResizableParallelRunnerCreate.argtypes = [vp]
        T.JxlResizableParallelRunnerDestroy.argtypes = [vp]
        T.JxlResizableParallelRunnerSetThreads.argtypes = [vp, C.c_size_t]
        T.JxlResizableParallelRunnerSuggestThreads.restype = C.c_uint32
        T.JxlResizableParallelRunnerSuggestThreads.argtypes = [C.c_uint64, C.c_uint64]
        _threads = T
    return _threads


def last_error() -> str:
    return libjxl().JxlHipLastError().decode()


# ---- errors (jpegxl-rs/src/errors.rs:27-52) ---------------------------------------------------------------------------
class DecodeError(Exception):
    pass


class CannotCreateDecoder(DecodeError):
    pass


class GenericError(DecodeError):
    pass


class InvalidInput(DecodeError):
    pass


class UnsupportedBitWidth(DecodeError):
    pass


class InternalError(DecodeError):
    pass


class UnknownStatus(DecodeError):
    pass


class NotImplementedFeature(DecodeError):
    pass


def check_dec_status(status: int):
    """errors.rs:93-99"""
    if status == JXL_DEC_SUCCESS:
        return
    if status == JXL_DEC_ERROR:
        raise GenericError(last_error())
    raise UnknownStatus(status)


def check_valid_signature(buf: bytes) -> Optional[bool]:
    """utils.rs:25-33: None = not enough bytes, False = invalid, True = codestream or container."""
    r = libjxl().JxlSignatureCheck(buf, len(buf))
    if r == 0:
        return None
    return r in (2, 3)


# ---- common (jpegxl-rs/src/common.rs) -----------------------------------------------------------------------------------
def icc_profile_from_headers(buf: bytes) -> bytes:
    """Extension (host-only, no GPU): the ICC profile JxlDecoderGetColorAsICCProfile reports for this file (decode.rs:368-385)."""
    L = libjxl()
    L.JxlHipColorProfileFromHeaders.argtypes = [C.c_char_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.JxlHipColorProfileFromHeaders.restype = C.c_int
    size = C.c_size_t(0)
    if L.JxlHipColorProfileFromHeaders(buf, len(buf), None, C.byref(size)):
        raise GenericError(last_error())
    out = (C.c_uint8 * size.value)()
    if L.JxlHipColorProfileFromHeaders(buf, len(buf), out, C.byref(size)):
        raise GenericError(last_error())
    return bytes(out)


class Endianness:
    Native, Little, Big = JXL_NATIVE_ENDIAN, JXL_LITTLE_ENDIAN, JXL_BIG_ENDIAN


_PIXEL_TYPES = {  # numpy dtype name -> (JxlDataType, bits, exponent bits)   common.rs:59-124
    "uint8": (JXL_TYPE_UINT8, 8, 0), "uint16": (JXL_TYPE_UINT16, 16, 0), "float32": (JXL_TYPE_FLOAT, 32, 8), "float16": (JXL_TYPE_FLOAT16, 16, 5)}


@dataclass
class PixelFormat:
    """decode.rs:49-82 — num_channels 0 = colour channels + alpha (if present)."""
    num_channels: int = 0
    endianness: int = Endianness.Native
    align: int = 0


@dataclass
class Metadata:
    """decode/result.rs:26-49"""
    width: int
    height: int
    intensity_target: float
    min_nits: float
    orientation: int
    num_color_channels: int
    has_alpha_channel: bool
    intrinsic_width: int
    intrinsic_height: int
    icc_profile: Optional[bytes] = None


# ---- parallel runners ---------------------------------------------------------------------------------------------------
class ThreadsRunner:
    """parallel/threads_runner.rs:30-89"""

    def __init__(self, num_workers: Optional[int] = None):
        T = libjxl_threads()
        n = T.JxlThreadParallelRunnerDefaultNumWorkerThreads() if num_workers is None else num_workers
        self._ptr = T.JxlThreadParallelRunnerCreate(None, n)
        if not self._ptr:
            raise CannotCreateDecoder("thread runner")

    def runner(self):
        return C.cast(libjxl_threads().JxlThreadParallelRunner, C.c_void_p)

    def as_opaque_ptr(self):
        return self._ptr

    def callback_basic_info(self, info):
        pass

    def __del__(self):
        if getattr(self, "_ptr", None):
            libjxl_threads().JxlThreadParallelRunnerDestroy(self._ptr)
            self._ptr = None


class ResizableRunner:
    """parallel/resizable_runner.rs:29-87"""

    def __init__(self):
        self._ptr = libjxl_threads().JxlResizableParallelRunnerCreate(None)
        if not self._ptr:
            raise CannotCreateDecoder("resizable runner")

    def runner(self):
        return C.cast(libjxl_threads().JxlResizableParallelRunner, C.c_void_p)

    def as_opaque_ptr(self):
        return self._ptr

    def callback_basic_info(self, info):  # resizable_runner.rs:78-80
        T = libjxl_threads()
        T.JxlResizableParallelRunnerSetThreads(self._ptr, T.JxlResizableParallelRunnerSuggestThreads(info.xsize, info.ysize))

    def __del__(self):
        if getattr(self, "_ptr", None):
            libjxl_threads().JxlResizableParallelRunnerDestroy(self._ptr)
            self._ptr = None


# ---- decoder (jpegxl-rs/src/decode.rs) ---------------------------------------------------------------------------------
class JxlDecoder:
    """Mirror of jpegxl_rs::decode::JxlDecoder.  All option fields are public and mutable after build (decode.rs:85-154)."""

    def __init__(self, pixel_format: Optional[PixelFormat] = None, skip_reorientation=None, unpremul_alpha=None,
                 render_spotcolors=None, coalescing=None, desired_intensity_target=None, decompress=None, progressive_detail=None,
                 icc_profile: bool = False, init_jpeg_buffer: int = 512 * 1024, parallel_runner=None, memory_manager=None):
        self.pixel_format = pixel_format
        self.skip_reorientation = skip_reorientation
        self.unpremul_alpha = unpremul_alpha
        self.render_spotcolors = render_spotcolors
        self.coalescing = coalescing
        self.desired_intensity_target = desired_intensity_target
        self.decompress = decompress                      # stored, never forwarded (decode.rs:129 vs 327-366)
        self.progressive_detail = progressive_detail      # ditto (decode.rs:135)
        self.icc_profile = icc_profile
        self.init_jpeg_buffer = init_jpeg_buffer
        self.parallel_runner = parallel_runner
        self.memory_manager = memory_manager
        L = libjxl()
        self._dec = L.JxlDecoderCreate(C.byref(memory_manager) if memory_manager is not None else None)  # decode.rs:177-182
        if not self._dec:
            raise CannotCreateDecoder()

    def __del__(self):
        if getattr(self, "_dec", None):
            libjxl().JxlDecoderDestroy(self._dec)  # decode.rs:519
            self._dec = None

    # decode.rs:327-366
    def _setup_decoder(self, icc: bool, reconstruct_jpeg: bool):
        L = libjxl()
        if self.parallel_runner is not None:
            check_dec_status(L.JxlDecoderSetParallelRunner(self._dec, self.parallel_runner.runner(), self.parallel_runner.as_opaque_ptr()))
        events = JXL_DEC_BASIC_INFO | JXL_DEC_FULL_IMAGE
        if icc:
            events |= JXL_DEC_COLOR_ENCODING
        if reconstruct_jpeg:
            events |= JXL_DEC_JPEG_RECONSTRUCTION
        check_dec_status(L.JxlDecoderSubscribeEvents(self._dec, events))
        if self.skip_reorientation is not None:
            check_dec_status(L.JxlDecoderSetKeepOrientation(self._dec, int(self.skip_reorientation)))
        if self.unpremul_alpha is not None:
            check_dec_status(L.JxlDecoderSetUnpremultiplyAlpha(self._dec, int(self.unpremul_alpha)))
        if self.render_spotcolors is not None:
            check_dec_status(L.JxlDecoderSetRenderSpotcolors(self._dec, int(self.render_spotcolors)))
        if self.coalescing is not None:
            check_dec_status(L.JxlDecoderSetCoalescing(self._dec, int(self.coalescing)))
        if self.desired_intensity_target is not None:
            check_dec_status(L.JxlDecoderSetDesiredIntensityTarget(self._dec, float(self.desired_intensity_target)))

    # decode.rs:368-385
    def _get_icc_profile(self) -> bytes:
        L = libjxl()
        size = C.c_size_t()
        check_dec_status(L.JxlDecoderGetICCProfileSize(self._dec, 1, C.byref(size)))
        buf = (C.c_uint8 * size.value)()
        check_dec_status(L.JxlDecoderGetColorAsICCProfile(self._dec, 1, buf, size.value))
        return bytes(buf)

    # decode.rs:387-434
    def _output(self, info: JxlBasicInfo, data_type: Optional[int], fmt: JxlPixelFormat):
        L = libjxl()
        if data_type is None:
            bits, exp = info.bits_per_sample, info.exponent_bits_per_sample
            if exp > 0:
                if bits == 16:
                    data_type = JXL_TYPE_FLOAT16
                elif bits == 32:
                    data_type = JXL_TYPE_FLOAT
                else:
                    raise UnsupportedBitWidth(bits)
            elif bits <= 8:
                data_type = JXL_TYPE_UINT8
            elif bits <= 16:
                data_type = JXL_TYPE_UINT16
            else:
                raise UnsupportedBitWidth(bits)
        f = self.pixel_format or PixelFormat()
        nc = f.num_channels if f.num_channels else info.num_color_channels + (1 if info.alpha_bits > 0 else 0)
        fmt.num_channels, fmt.data_type, fmt.endianness, fmt.align = nc, data_type, f.endianness, f.align
        size = C.c_size_t()
        check_dec_status(L.JxlDecoderImageOutBufferSize(self._dec, C.byref(fmt), C.byref(size)))
        pixels = np.zeros(size.value, dtype=np.uint8)  # Vec::resize(size, 0)
        check_dec_status(L.JxlDecoderSetImageOutBuffer(self._dec, C.byref(fmt), pixels.ctypes.data, size.value))
        return pixels

    # decode.rs:207-325
    def _decode_internal(self, data: bytes, data_type: Optional[int], with_icc: bool, reconstruct: bool):
        L = libjxl()
        sig = check_valid_signature(data)
        if sig is None or not sig:
            raise InvalidInput()
        self._setup_decoder(with_icc, reconstruct)
        inbuf = np.frombuffer(data, dtype=np.uint8)
        check_dec_status(L.JxlDecoderSetInput(self._dec, inbuf.ctypes.data, len(data)))
        L.JxlDecoderCloseInput(self._dec)
        info = JxlBasicInfo()
        fmt = JxlPixelFormat()
        pixels = np.zeros(0, dtype=np.uint8)
        icc = None
        jpeg = None
        while True:
            status = L.JxlDecoderProcessInput(self._dec)
            if status in (JXL_DEC_NEED_MORE_INPUT, JXL_DEC_ERROR):
                raise GenericError(last_error())          # NB: no Reset on error paths (decode.rs:241)
            elif status == JXL_DEC_BASIC_INFO:
                check_dec_status(L.JxlDecoderGetBasicInfo(self._dec, C.byref(info)))
                if self.parallel_runner is not None:
                    self.parallel_runner.callback_basic_info(info)
            elif status == JXL_DEC_COLOR_ENCODING:
                icc = self._get_icc_profile()
            elif status == JXL_DEC_JPEG_RECONSTRUCTION:
                jpeg = np.zeros(self.init_jpeg_buffer, dtype=np.uint8)
                check_dec_status(L.JxlDecoderSetJPEGBuffer(self._dec, jpeg.ctypes.data, len(jpeg)))
            elif status == JXL_DEC_JPEG_NEED_MORE_OUTPUT:
                need = L.JxlDecoderReleaseJPEGBuffer(self._dec)
                jpeg = np.concatenate([jpeg, np.zeros(need, dtype=np.uint8)])
                check_dec_status(L.JxlDecoderSetJPEGBuffer(self._dec, jpeg.ctypes.data, len(jpeg)))
            elif status == JXL_DEC_NEED_IMAGE_OUT_BUFFER:
                pixels = self._output(info, data_type, fmt)
            elif status in (JXL_DEC_FULL_IMAGE, JXL_DEC_FRAME, JXL_DEC_FRAME_PROGRESSION):
                continue
            elif status == JXL_DEC_SUCCESS:
                if jpeg is not None:
                    remaining = L.JxlDecoderReleaseJPEGBuffer(self._dec)
                    jpeg = jpeg[: len(jpeg) - remaining]
                L.JxlDecoderReset(self._dec)
                meta = Metadata(info.xsize, info.ysize, info.intensity_target, info.min_nits, info.orientation, info.num_color_channels,
                                info.alpha_bits > 0, info.intrinsic_xsize, info.intrinsic_ysize, icc)
                return meta, fmt, pixels, jpeg
            elif status == JXL_DEC_NEED_PREVIEW_OUT_BUFFER:
                raise NotImplementedFeature("preview image output")
            elif status == JXL_DEC_BOX_NEED_MORE_OUTPUT:
                raise NotImplementedFeature("box output")
            elif status == JXL_DEC_PREVIEW_IMAGE:
                raise NotImplementedFeature("preview image")
            elif status == JXL_DEC_BOX:
                raise NotImplementedFeature("box handling")
            elif status == JXL_DEC_BOX_COMPLETE:
                raise NotImplementedFeature("box complete")
            else:
                raise UnknownStatus(status)

    @staticmethod
    def _convert(raw: np.ndarray, fmt: JxlPixelFormat) -> np.ndarray:
        """PixelType::convert (common.rs:59-124): endian-aware reinterpretation of the byte buffer."""
        dt = {JXL_TYPE_UINT8: "u1", JXL_TYPE_UINT16: "u2", JXL_TYPE_FLOAT: "f4", JXL_TYPE_FLOAT16: "f2"}[fmt.data_type]
        order = ">" if fmt.endianness == JXL_BIG_ENDIAN else "<"
        n = len(raw) // np.dtype(dt).itemsize
        return raw[: n * np.dtype(dt).itemsize].view(np.dtype(order + dt)).astype(np.dtype(dt))

    def decode(self, data: bytes):
        """decode.rs:440-455 — returns (Metadata, pixels) with the sample type inferred from the header."""
        meta, fmt, pixels, _ = self._decode_internal(data, None, self.icc_profile, False)
        return meta, self._convert(pixels, fmt)

    def decode_with(self, data: bytes, dtype):
        """decode.rs:461-484 — decode_with::<T>; dtype in {uint8, uint16, float32, float16}."""
        name = np.dtype(dtype).name
        if name not in _PIXEL_TYPES:
            raise UnsupportedBitWidth(name)
        meta, fmt, pixels, _ = self._decode_internal(data, _PIXEL_TYPES[name][0], self.icc_profile, False)
        return meta, self._convert(pixels, fmt)

    # ---- image.rs:32-132, the `image` crate integration (trait ToDynamic): the decoded buffer as an image object, None when the combination
    # of sample type and channel count has no DynamicImage variant.  The stand-in for DynamicImage here is a (height, width, channels) numpy
    # array tagged with the variant's name.
    _DYNAMIC = {("float32", 3): "ImageRgb32F", ("float32", 4): "ImageRgba32F", ("uint8", 1): "ImageLuma8", ("uint8", 2): "ImageLumaA8",
                ("uint8", 3): "ImageRgb8", ("uint8", 4): "ImageRgba8", ("uint16", 1): "ImageLuma16", ("uint16", 2): "ImageLumaA16",
                ("uint16", 3): "ImageRgb16", ("uint16", 4): "ImageRgba16"}

    def _to_image(self, meta, fmt, pixels):
        arr = self._convert(pixels, fmt)
        variant = self._DYNAMIC.get((arr.dtype.name, fmt.num_channels))
        if variant is None or arr.size != meta.width * meta.height * fmt.num_channels:     # (ImageBuffer::from_raw: the buffer must hold the image exactly)
            return None
        return DynamicImage(variant, arr.reshape(meta.height, meta.width, fmt.num_channels))

    def decode_to_image(self, data: bytes):
        """image.rs:53-66 — decode with the sample type the header asks for; Optional[DynamicImage]."""
        meta, fmt, pixels, _ = self._decode_internal(data, None, False, False)
        return self._to_image(meta, fmt, pixels)

    def decode_to_image_with(self, data: bytes, dtype):
        """image.rs:68-86 — decode_to_image_with::<T>."""
        name = np.dtype(dtype).name
        if name not in _PIXEL_TYPES:
            raise UnsupportedBitWidth(name)
        meta, fmt, pixels, _ = self._decode_internal(data, _PIXEL_TYPES[name][0], False, False)
        return self._to_image(meta, fmt, pixels)

    def reconstruct(self, data: bytes):
        """decode.rs:493-514 — returns (Metadata, ('jpeg', bytes) | ('pixels', ndarray))."""
        meta, fmt, pixels, jpeg = self._decode_internal(data, None, self.icc_profile, True)
        if jpeg is not None and len(jpeg):
            return meta, ("jpeg", jpeg.tobytes())
        return meta, ("pixels", self._convert(pixels, fmt))


class DynamicImage:
    """Stand-in for image::DynamicImage (image.rs:88-132): `variant` names the enum variant, `pixels` is (height, width, channels)."""

    def __init__(self, variant: str, pixels: np.ndarray):
        self.variant, self.pixels = variant, pixels

    def to_rgba16(self) -> np.ndarray:
        """DynamicImage::to_rgba16 as the reference's test uses it (image.rs:169): grey is replicated, alpha defaults to opaque, 8-bit samples
        scale by 257, float samples by 65535 (clamped)."""
        p = self.pixels
        if p.dtype == np.uint8:
            p = p.astype(np.uint16) * 257
        elif p.dtype == np.float32:
            p = np.rint(np.clip(p, 0.0, 1.0) * 65535.0).astype(np.uint16)
        c = p.shape[2]
        rgb = np.repeat(p[:, :, :1], 3, axis=2) if c <= 2 else p[:, :, :3]
        alpha = p[:, :, c - 1:c] if c in (2, 4) else np.full(p.shape[:2] + (1,), 65535, np.uint16)
        return np.concatenate([rgb, alpha], axis=2).astype(np.uint16)


def decoder_builder(**options) -> JxlDecoder:
    """jpegxl-rs/src/decode.rs:535 — `decoder_builder().pixel_format(..).build()` becomes keyword arguments."""
    return JxlDecoder(**options)


# ---- batch extension (include/jxl_hip.h, JxlHipBatch*) -------------------------------------------------------------------
class BatchDecoder:
    """Device-resident decode of a batch of independent images: inputs and outputs stay in HBM."""

    def __init__(self, device: int = 0):
        L = libjxl()
        self._h = L.JxlHipBatchCreate(device)
        if not self._h:
            raise CannotCreateDecoder(last_error())
        self._n = 0
        self._fmt = []

    def __del__(self):
        if getattr(self, "_h", None):
            libjxl().JxlHipBatchDestroy(self._h)
            self._h = None

    def _chk(self, status):
        if status != JXL_DEC_SUCCESS:
            raise GenericError(last_error())

    def add(self, data: bytes, dtype="uint8", num_channels=0, endianness=Endianness.Native, align=0, device_ptr=None) -> int:
        L = libjxl()
        buf = np.frombuffer(data, dtype=np.uint8)
        i = L.JxlHipBatchAddImage(self._h, buf.ctypes.data, len(data))
        if i < 0:
            raise GenericError(last_error())
        fmt = JxlPixelFormat(num_channels, _PIXEL_TYPES[np.dtype(dtype).name][0], endianness, align)
        self._chk(L.JxlHipBatchSetOutput(self._h, i, C.byref(fmt), device_ptr))
        self._fmt.append(fmt)
        self._n += 1
        return i

    def add_many(self, datas, dtype="uint8", num_channels=0, device_ptrs=None, threads=4, endianness=Endianness.Native, align=0) -> int:
        """Parses the images of `datas` (bytes objects) on `threads` host threads and appends them in order (JxlHipBatchAddImages);
        device_ptrs: optional caller-owned device destination per image.  Returns the index of the first one."""
        L = libjxl()
        n = len(datas)
        ptrs = (C.c_char_p * n)(*datas)
        sizes = (C.c_size_t * n)(*[len(d) for d in datas])
        first = L.JxlHipBatchAddImages(self._h, ptrs, sizes, n, int(threads))
        if first < 0:
            raise GenericError(last_error())
        fmt = JxlPixelFormat(num_channels, _PIXEL_TYPES[np.dtype(dtype).name][0], endianness, align)
        for k in range(n):
            self._chk(L.JxlHipBatchSetOutput(self._h, first + k, C.byref(fmt), device_ptrs[k] if device_ptrs is not None else None))
            self._fmt.append(fmt)
        self._n += n
        return first

    def reset(self):
        """Forgets the images, keeps the device arenas and buffer sharing (JxlHipBatchReset): fill and prepare again."""
        libjxl().JxlHipBatchReset(self._h)
        self._n = 0
        self._fmt = []

    def info(self, i) -> JxlBasicInfo:
        info = JxlBasicInfo()
        self._chk(libjxl().JxlHipBatchGetBasicInfo(self._h, i, C.byref(info)))
        return info

    def out_size(self, i) -> int:
        s = C.c_size_t()
        self._chk(libjxl().JxlHipBatchOutBufferSize(self._h, i, C.byref(self._fmt[i]), C.byref(s)))
        return s.value

    def set_lane_stride(self, lf=64, hf=64):
        libjxl().JxlHipBatchSetLaneStride(self._h, lf, hf)

    def set_option(self, name: str, value: int):
        libjxl().JxlHipBatchSetOption(self._h, name.encode(), int(value))

    def share_buffers(self, owner: "BatchDecoder"):
        """Use `owner`'s coefficient / pixel planes (call before prepare; see include/jxl_hip.h JxlHipBatchShareBuffers)."""
        self._chk(libjxl().JxlHipBatchShareBuffers(self._h, owner._h if owner is not None else None))   # (None: planes of its own again)
        self._owner = owner   # keep it alive

    def share_coefficients(self, owner: "BatchDecoder"):
        """Use `owner`'s quantised-coefficient planes (call before prepare; include/jxl_hip.h JxlHipBatchShareCoefficients)."""
        self._chk(libjxl().JxlHipBatchShareCoefficients(self._h, owner._h if owner is not None else None))
        self._coef_owner = owner

    def prepare(self, stream=None):
        self._chk(libjxl().JxlHipBatchPrepare(self._h, stream))

    def decode(self, stream=None):
        self._chk(libjxl().JxlHipBatchDecode(self._h, stream))

    def decode_timed(self, stream=None):
        self._chk(libjxl().JxlHipBatchDecodeTimed(self._h, stream))

    def decode_part(self, part: int, stream=None, timed=False):
        """part 1 = front (LF stage), 2 = rest (HF, IDCT, filters, output) — or 3 = HF only, 4 = IDCT, filters, output; see include/jxl_hip.h."""
        self._chk(libjxl().JxlHipBatchDecodePart(self._h, stream, part, 1 if timed else 0))

    def collect_times(self):
        """-> (dict stage -> summed ms, number of timed decodes)"""
        t = JxlHipStageTimes(); runs = C.c_int()
        self._chk(libjxl().JxlHipBatchCollectTimes(self._h, C.byref(t), C.byref(runs)))
        return {n: getattr(t, n) for n, _ in JxlHipStageTimes._fields_}, runs.value

    def finish(self, stream=None):
        self._chk(libjxl().JxlHipBatchFinish(self._h, stream))

    def device_output(self, i) -> int:
        return libjxl().JxlHipBatchDeviceOutput(self._h, i)

    def output(self, i, stream=None) -> np.ndarray:
        n = self.out_size(i)
        out = np.empty(n, dtype=np.uint8)
        self._chk(libjxl().JxlHipBatchCopyOutput(self._h, i, out.ctypes.data, n, stream))
        return JxlDecoder._convert(out, self._fmt[i])

    def debug_read(self, i: int, name: str, channel: int = 0, dtype=np.float32) -> np.ndarray:
        """Testing: a device buffer of image i's first frame after a decode (include/jxl_hip.h JxlHipBatchDebugRead), as a flat array."""
        L = libjxl()
        n = L.JxlHipBatchDebugRead(self._h, i, name.encode(), channel, None, 0, None)
        if n == 0:
            raise GenericError(last_error())
        out = np.empty(n // np.dtype(dtype).itemsize, dtype=dtype)
        L.JxlHipBatchDebugRead(self._h, i, name.encode(), channel, out.ctypes.data, n, None)
        return out

    def info_value(self, name: str) -> int:
        """Facts about the prepared batch by name (include/jxl_hip.h JxlHipBatchGetInfo), e.g. "lf_simt_frames"."""
        return int(libjxl().JxlHipBatchGetInfo(self._h, name.encode()))

    @property
    def total_pixels(self):
        return libjxl().JxlHipBatchTotalPixels(self._h)

    @property
    def compressed_bytes(self):
        return libjxl().JxlHipBatchCompressedBytes(self._h)

    @property
    def stage_bytes(self):
        """Algorithmic HBM bytes of one decode of the batch per stage (lf, lfpost, hf, idct, filters, out)."""
        a = (C.c_uint64 * 6)()
        libjxl().JxlHipBatchStageBytes(self._h, C.byref(a))
        return dict(zip(("lf", "lfpost", "hf", "idct", "filter", "out"), [int(v) for v in a]))

    @property
    def device_bytes(self):
        return libjxl().JxlHipBatchDeviceBytes(self._h)


def arena_pool_trim() -> int:
    """Hands every pooled device arena back to the HIP runtime (JxlHipArenaPoolTrim); returns the bytes released."""
    return int(libjxl().JxlHipArenaPoolTrim())


def image_out_size(data: bytes, dtype="uint8", num_channels=0, endianness=Endianness.Native, align=0):
    """(JxlBasicInfo, bytes of the decoded image in that format) from the headers alone — host-only (JxlHipImageOutSize)."""
    fmt = JxlPixelFormat(num_channels, _PIXEL_TYPES[np.dtype(dtype).name][0], endianness, align)
    info, size = JxlBasicInfo(), C.c_size_t()
    buf = np.frombuffer(data, dtype=np.uint8)
    check_dec_status(libjxl().JxlHipImageOutSize(buf.ctypes.data, len(data), C.byref(fmt), C.byref(info), C.byref(size)))
    return info, size.value


class PinnedBuffer:
    """Pinned host memory (JxlHipHostAlloc) as a numpy uint8 array: destination of a pipeline's host_out copies."""

    def __init__(self, nbytes: int):
        self._p = libjxl().JxlHipHostAlloc(max(1, int(nbytes)))
        if not self._p:
            raise MemoryError(last_error())
        self.nbytes = int(nbytes)
        self.array = np.ctypeslib.as_array((C.c_uint8 * max(1, self.nbytes)).from_address(self._p))[: self.nbytes]

    @property
    def ptr(self) -> int:
        return self._p

    def __del__(self):
        if getattr(self, "_p", None):
            self.array = None
            libjxl().JxlHipHostFree(self._p)
            self._p = None


class Pipeline:
    """The streaming decode pipeline of one GPU (include/jxl_hip.h JxlHipPipeline*; csrc/pipeline.h): submit jobs of compressed images, the library overlaps
    their stages — host parse / upload, LF, HF, IDCT, filters, copies — over its own streams and threads.  submit() returns a ticket, wait(ticket) the per-image
    status once the pixels are where they were asked for."""

    def __init__(self, device: int = 0, **options):
        o = JxlHipPipelineOptions()
        o.wide_first = -1
        for k, v in options.items():
            if not hasattr(o, k):
                raise TypeError(f"unknown pipeline option {k}")
            setattr(o, k, int(v))
        self._h = libjxl().JxlHipPipelineCreate(int(device), C.byref(o))
        if not self._h:
            raise CannotCreateDecoder(last_error())
        self._keep = {}

    def close(self):
        if getattr(self, "_h", None):
            libjxl().JxlHipPipelineDestroy(self._h)
            self._h = None
            self._keep = {}

    __del__ = close

    def submit(self, datas, dtype="uint8", num_channels=0, device_ptrs=None, host_ptrs=None, capacities=None, endianness=Endianness.Native, align=0) -> int:
        """Job of len(datas) images (bytes objects).  device_ptrs / host_ptrs: one destination address per image (exactly one of the two lists); the bytes objects and
        the destinations are kept referenced until wait()."""
        n = len(datas)
        ptrs = (C.c_char_p * n)(*datas)
        sizes = (C.c_size_t * n)(*[len(d) for d in datas])
        fmt = JxlPixelFormat(num_channels, _PIXEL_TYPES[np.dtype(dtype).name][0], endianness, align)
        dev = (C.c_void_p * n)(*device_ptrs) if device_ptrs is not None else None
        host = (C.c_void_p * n)(*host_ptrs) if host_ptrs is not None else None
        caps = (C.c_size_t * n)(*capacities) if capacities is not None else None
        t = libjxl().JxlHipPipelineSubmit(self._h, ptrs, sizes, n, C.byref(fmt), dev, host, caps)
        if t < 0:
            raise GenericError(last_error())
        self._keep[t] = (datas, ptrs, sizes, dev, host, caps, n)
        return t

    def wait(self, ticket: int, check=True):
        """-> (per-image status list, end_ms).  check: raise GenericError if an image failed."""
        n = self._keep[ticket][6] if ticket in self._keep else 0
        st = (C.c_int * max(1, n))()
        end = C.c_float()
        rc = libjxl().JxlHipPipelineWait(self._h, int(ticket), st, n, C.byref(end))
        self._keep.pop(ticket, None)
        if rc != JXL_DEC_SUCCESS and check:
            raise GenericError(last_error())
        return list(st)[:n], float(end.value)

    def wait_all(self):
        check_dec_status(libjxl().JxlHipPipelineWaitAll(self._h))

    def reset_clock(self):
        check_dec_status(libjxl().JxlHipPipelineResetClock(self._h))

    def collect_times(self):
        t, runs = JxlHipStageTimes(), C.c_int()
        check_dec_status(libjxl().JxlHipPipelineCollectTimes(self._h, C.byref(t), C.byref(runs)))
        return {n: getattr(t, n) for n, _ in JxlHipStageTimes._fields_}, runs.value

    @property
    def stage_bytes(self):
        out = (C.c_uint64 * 6)()
        libjxl().JxlHipPipelineStageBytes(self._h, C.byref(out))
        return dict(zip(("lf", "lfpost", "hf", "idct", "filter", "out"), [int(v) for v in out]))

    def info(self, name: str) -> int:
        return int(libjxl().JxlHipPipelineGetInfo(self._h, name.encode()))
