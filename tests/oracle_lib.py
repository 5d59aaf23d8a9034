"""ctypes binding of the CPU oracle (oracle/_build/libjxl_oracle.so).  TEST INFRASTRUCTURE ONLY."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "libjxl_oracle.so")


class Info(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in (
        "xsize", "ysize", "bits_per_sample", "exponent_bits", "num_color_channels", "num_extra_channels", "alpha_bits",
        "orientation", "have_container", "xyb_encoded", "has_jbrd", "reserved")] + [
        ("intensity_target", C.c_float), ("min_nits", C.c_float),
        ("tokens_lf", C.c_uint64), ("tokens_hf", C.c_uint64), ("tokens_modular", C.c_uint64), ("seconds", C.c_double)]


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.jxlo_decode.restype = C.c_void_p
        L.jxlo_decode.argtypes = [C.c_char_p, C.c_size_t, C.c_int]
        L.jxlo_error.restype = C.c_char_p
        L.jxlo_error.argtypes = [C.c_void_p]
        L.jxlo_free.argtypes = [C.c_void_p]
        L.jxlo_get_info.argtypes = [C.c_void_p, C.POINTER(Info)]
        L.jxlo_render.restype = C.c_size_t
        L.jxlo_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_size_t, C.POINTER(C.POINTER(C.c_uint8))]
        L.jxlo_get_plane.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.jxlo_get_ints.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.POINTER(C.c_int32)), C.POINTER(C.c_size_t)]
        L.jxlo_idct.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        L.jxlo_natural_order.argtypes = [C.c_int, C.c_void_p]
        L.jxlo_srgb.restype = C.c_float
        L.jxlo_srgb.argtypes = [C.c_float]
        _lib = L
    return _lib


class OracleError(RuntimeError):
    pass


TYPES = {"u8": (0, np.uint8), "u16": (1, np.uint16), "f32": (2, np.float32), "f16": (3, np.float16)}


class Decoded:
    """One decoded image held by the oracle."""

    def __init__(self, data: bytes, dump: bool = False):
        self._L = lib()
        self._h = self._L.jxlo_decode(data, len(data), 1 if dump else 0)
        err = self._L.jxlo_error(self._h)
        if err:
            msg = err.decode()
            self.close()
            raise OracleError(msg)
        self.info = Info()
        self._L.jxlo_get_info(self._h, C.byref(self.info))

    def close(self):
        if self._h:
            self._L.jxlo_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def set_unpremultiply_alpha(self, on=True):
        """JxlDecoderSetUnpremultiplyAlpha: colour / alpha in the write stage when the alpha channel is associated."""
        self._L.jxlo_set_unpremultiply_alpha.argtypes = [C.c_void_p, C.c_int]
        self._L.jxlo_set_unpremultiply_alpha(self._h, 1 if on else 0)

    def icc(self) -> bytes:
        """The embedded ICC profile (b"" when the colour encoding is enumerated)."""
        self._L.jxlo_icc.restype = C.c_size_t
        self._L.jxlo_icc.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        n = self._L.jxlo_icc(self._h, None, 0)
        buf = C.create_string_buffer(n)
        self._L.jxlo_icc(self._h, buf, n)
        return buf.raw[:n]

    def pixels(self, dtype="u8", num_channels=0, big_endian=False, align=0):
        t, npdt = TYPES[dtype]
        if num_channels == 0:
            num_channels = self.info.num_color_channels + (1 if self.info.alpha_bits else 0)
        p = C.POINTER(C.c_uint8)()
        n = self._L.jxlo_render(self._h, t, num_channels, 1 if big_endian else 0, align, C.byref(p))
        buf = np.ctypeslib.as_array(p, shape=(n,)).copy()
        return buf

    def image(self, dtype="u8", num_channels=0):
        t, npdt = TYPES[dtype]
        if num_channels == 0:
            num_channels = self.info.num_color_channels + (1 if self.info.alpha_bits else 0)
        raw = self.pixels(dtype, num_channels)
        return raw.view(npdt).reshape(self.info.ysize, self.info.xsize, num_channels)

    def plane(self, name):
        p = C.POINTER(C.c_float)(); w = C.c_int(); h = C.c_int()
        if not self._L.jxlo_get_plane(self._h, name.encode(), C.byref(p), C.byref(w), C.byref(h)):
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(h.value, w.value)).copy()

    def ints(self, name):
        p = C.POINTER(C.c_int32)(); n = C.c_size_t()
        if not self._L.jxlo_get_ints(self._h, name.encode(), C.byref(p), C.byref(n)):
            raise KeyError(name)
        return np.ctypeslib.as_array(p, shape=(n.value,)).copy()


def decode(data: bytes, dump=False, dc_only=False, max_passes=-1, allow_truncated=False) -> Decoded:
    """dc_only: the image as JxlDecoderFlushImage shows it at the kDC progression step — LF image + HF metadata decoded, every AC coefficient zero; the
    PassGroup sections are not read (the input may end in them).  max_passes: a later progression step — only that many passes of every group.
    allow_truncated: `data` may end inside the frame's PassGroup sections; groups are drawn with the passes that are completely there."""
    if max_passes >= 0 or allow_truncated:
        L = lib()
        L.jxlo_set_progress.argtypes = [C.c_int, C.c_int]
        L.jxlo_set_progress(max_passes, 1 if allow_truncated else 0)
        try:
            return Decoded(data, dump)
        finally:
            L.jxlo_set_progress(-1, 0)
    if dc_only:
        L = lib()
        L.jxlo_set_dc_only.argtypes = [C.c_int]
        L.jxlo_set_dc_only(1)
        try:
            return Decoded(data, dump)
        finally:
            L.jxlo_set_dc_only(0)
    return Decoded(data, dump)


def set_render_spotcolors(on=True):
    """JxlDecoderSetRenderSpotcolors for the oracle decodes started afterwards (default on)."""
    L = lib()
    L.jxlo_set_render_spotcolors.argtypes = [C.c_int]
    L.jxlo_set_render_spotcolors(1 if on else 0)
