"""ctypes binding of tools/_build/libjxlsynth.so — deterministic JPEG XL bit-stream synthesiser (fixture generator)."""
import ctypes as C
import os
import subprocess
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOLS_DIR = os.path.join(ROOT, "tools")
LIB_PATH = os.path.join(TOOLS_DIR, "_build", "libjxlsynth.so")


class Params(C.Structure):
    _fields_ = [("seed", C.c_uint32), ("distance", C.c_float), ("epf_iters", C.c_int32), ("gab", C.c_int32),
                ("strategy_mix", C.c_int32), ("out_bits", C.c_int32), ("hdr", C.c_int32), ("skip_lf_smoothing", C.c_int32),
                ("custom_orders", C.c_int32), ("orientation", C.c_int32), ("upsampling", C.c_int32), ("custom_up_weights", C.c_int32),
                ("num_passes", C.c_int32), ("permute_toc", C.c_int32), ("pass_ds", C.c_int32), ("reserved", C.c_int32 * 2)]


class Frame(C.Structure):
    """Frame control of the synthesiser (tools/jxl_synth.cc jxlsynth_frame)."""
    _fields_ = [("noise", C.c_int32), ("noise_lut", C.c_uint32 * 8)] + [(n, C.c_int32) for n in (
        "frame_type", "have_crop", "crop_x0", "crop_y0", "canvas_w", "canvas_h", "blend_mode", "blend_source", "blend_clamp", "is_last",
        "save_as_reference", "save_before_ct", "emit", "num_extra_hdr", "xyb_image", "alpha_premultiplied", "use_lf_frame", "lf_level", "mod_passes", "mod_ds", "duration")]


def frame(**kw):
    f = Frame(is_last=1, num_extra_hdr=-1, mod_passes=1, mod_ds=1)
    lut = kw.pop("noise_lut", None)
    if lut is not None:
        f.noise = 1
        for i, v in enumerate(lut):
            f.noise_lut[i] = int(v)
    for k, v in kw.items():
        setattr(f, k, int(v))
    return f


_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", TOOLS_DIR])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.jxlsynth_last_error.restype = C.c_char_p
        L.jxlsynth_free.argtypes = [C.c_void_p]
        L.jxlsynth_image.argtypes = [C.c_uint32, C.c_int, C.c_int, C.c_void_p]
        L.jxlsynth_vardct.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.jxlsynth_vardct2.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.jxlsynth_modular.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.jxlsynth_modular2.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.jxlsynth_vardct3.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(Params), C.POINTER(Frame), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.jxlsynth_modular3.argtypes = [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(Frame), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _lib = L
    return _lib


def synthetic_image(seed, w, h):
    """Seeded synthetic sRGB u8 image: low-frequency cosines + soft rectangles + noise."""
    out = np.empty((h, w, 3), np.uint8)
    lib().jxlsynth_image(seed, w, h, out.ctypes.data)
    return out


def _take(out, n):
    data = C.string_at(out.value, n.value)
    lib().jxlsynth_free(out)
    return data


def encode_vardct(rgb, seed=1, distance=1.0, epf_iters=1, gab=1, strategy_mix=1, out_bits=8, hdr=0, skip_lf_smoothing=0, orientation=1, alpha=None, upsampling=1, custom_up_weights=0, num_passes=1, permute_toc=0, pass_ds=0):
    """rgb: (h,w,3) uint8 sRGB, or float32 linear when hdr=1; alpha: optional (h,w) uint8 plane carried as a lossless
    extra channel.  Returns codestream bytes."""
    L = lib()
    h, w = rgb.shape[:2]
    p = Params(seed=seed, distance=distance, epf_iters=epf_iters, gab=gab, strategy_mix=strategy_mix, out_bits=out_bits,
               hdr=hdr, skip_lf_smoothing=skip_lf_smoothing, orientation=orientation, upsampling=upsampling, custom_up_weights=custom_up_weights, num_passes=num_passes, permute_toc=permute_toc, pass_ds=pass_ds)
    out = C.c_void_p(); n = C.c_size_t()
    alpha_arr = None if alpha is None else np.ascontiguousarray(alpha, dtype=np.uint8)
    al = None if alpha_arr is None else alpha_arr.ctypes.data
    if rgb.dtype == np.uint8:
        a = np.ascontiguousarray(rgb)
        rc = L.jxlsynth_vardct2(a.ctypes.data, None, al, w, h, C.byref(p), C.byref(out), C.byref(n))
    else:
        a = np.ascontiguousarray(rgb, dtype=np.float32)
        rc = L.jxlsynth_vardct2(None, a.ctypes.data, al, w, h, C.byref(p), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


def encode_modular(img, bits=8, rct=False, squeeze=0):
    """img: (h,w,C) integer array, C in {1,2,3,4} (2/4 = with alpha).  Lossless Modular codestream.
    squeeze: 0 none, 1 default Squeeze chain (what `cjxl -d 0 -R 1` signals), 2 short explicit chain."""
    L = lib()
    h, w, c = img.shape
    has_alpha = c in (2, 4)
    nchan = c - (1 if has_alpha else 0)
    planes = [np.ascontiguousarray(img[..., i].astype(np.int32)) for i in range(c)]
    arr = (C.c_void_p * c)(*[p.ctypes.data for p in planes])
    out = C.c_void_p(); n = C.c_size_t()
    rc = L.jxlsynth_modular2(arr, nchan, 1 if has_alpha else 0, w, h, bits, 1 if rct else 0, int(squeeze), C.byref(out), C.byref(n))
    if rc:
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


def encode_vardct_frame(rgb, fx, seed=1, distance=1.0, epf_iters=1, gab=1, strategy_mix=1, alpha=None):
    """One VarDCT frame under frame control `fx` (see frame()): emit 0 = image header + frame, 1 = frame only."""
    L = lib()
    h, w = rgb.shape[:2]
    p = Params(seed=seed, distance=distance, epf_iters=epf_iters, gab=gab, strategy_mix=strategy_mix, out_bits=8, orientation=1, upsampling=1, num_passes=1)
    out = C.c_void_p(); n = C.c_size_t()
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    alpha_arr = None if alpha is None else np.ascontiguousarray(alpha, dtype=np.uint8)
    if L.jxlsynth_vardct3(a.ctypes.data, None if alpha_arr is None else alpha_arr.ctypes.data, w, h, C.byref(p), C.byref(fx), C.byref(out), C.byref(n)):
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


def encode_modular_frame(img, fx, bits=8, rct=False, squeeze=0):
    """One lossless Modular frame under frame control `fx`."""
    L = lib()
    h, w, c = img.shape
    has_alpha = c in (2, 4)
    nchan = c - (1 if has_alpha else 0)
    planes = [np.ascontiguousarray(img[..., i].astype(np.int32)) for i in range(c)]
    arr = (C.c_void_p * c)(*[p.ctypes.data for p in planes])
    out = C.c_void_p(); n = C.c_size_t()
    if L.jxlsynth_modular3(arr, nchan, 1 if has_alpha else 0, w, h, bits, 1 if rct else 0, int(squeeze), C.byref(fx), C.byref(out), C.byref(n)):
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


class FreeParams(C.Structure):
    """tools/jxl_synth.cc jxlsynth_free_params (tools/synth_free.h FreeParams)."""
    _fields_ = [("seed", C.c_uint32)] + [(n, C.c_int) for n in ("w", "h", "nchan", "has_alpha", "bits", "tree_flags", "tree_depth", "local_trees", "lz77",
                                                                 "palette", "nb_colors", "nb_deltas", "pal_pred")]


TREE_WP, TREE_PREV_CHANNELS, TREE_MULTIPLIERS, TREE_ALL_PREDICTORS, TREE_CUSTOM_WP = 1, 2, 4, 8, 16


def encode_modular_free(seed=1, w=64, h=64, nchan=3, has_alpha=False, bits=8, tree_flags=0, tree_depth=5, local_trees=0, lz77=False,
                        palette=False, nb_colors=16, nb_deltas=0, pal_pred=0):
    """Free-running Modular stream (tools/synth_free.h): random MA trees / predictors / properties, local trees, LZ77, delta
    palettes.  The pixels are whatever a decoder makes of the token stream — for decoder-vs-decoder parity only."""
    L = lib()
    L.jxlsynth_modular_free.argtypes = [C.POINTER(FreeParams), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    p = FreeParams(seed, w, h, nchan, 1 if has_alpha else 0, bits, tree_flags, tree_depth, local_trees, 1 if lz77 else 0,
                   1 if palette else 0, nb_colors, nb_deltas, pal_pred)
    out = C.c_void_p(); n = C.c_size_t()
    if L.jxlsynth_modular_free(C.byref(p), C.byref(out), C.byref(n)):
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


def set_icc(icc: bytes = b""):
    """Embed `icc` (an ICC profile) in the image headers written from now on (b"" = back to enumerated colour encodings)."""
    L = lib()
    L.jxlsynth_set_icc.argtypes = [C.c_char_p, C.c_size_t]
    L.jxlsynth_set_icc(icc, len(icc))


def set_features(patches=None, splines=None, num_extra=0):
    """Patch dictionary / splines of the frames written from now on (tools/jxl_synth.cc WriteFeatures); call without arguments to clear.
    patches: list of (ref, x0, y0, xsize, ysize, [(x, y, [(mode, alpha_channel, clamp)] * (1 + num_extra)), ...])
    splines: (quant_adjust, [(start_x, start_y, [(ddx, ddy), ...], colour_dct[3][32], sigma_dct[32]), ...])"""
    L = lib()
    L.jxlsynth_set_features.argtypes = [C.POINTER(C.c_int32), C.c_size_t, C.POINTER(C.c_int32), C.c_size_t, C.c_int]
    pf, sf = [], []
    for ref, x0, y0, xs, ys, positions in (patches or []):
        pf += [ref, x0, y0, xs, ys, len(positions)]
        for x, y, blends in positions:
            assert len(blends) == 1 + num_extra
            pf += [x, y]
            for mode, alpha, clamp in blends:
                pf += [mode, alpha, clamp]
    if splines:
        adjust, items = splines
        sf += [adjust, len(items)]
        for sx, sy, cps, colour, sigma in items:
            sf += [sx, sy, len(cps)]
            for dx, dy in cps:
                sf += [dx, dy]
            for c in range(3):
                assert len(colour[c]) == 32
                sf += [int(v) for v in colour[c]]
            assert len(sigma) == 32
            sf += [int(v) for v in sigma]
    pa = (C.c_int32 * max(1, len(pf)))(*pf)
    sa = (C.c_int32 * max(1, len(sf)))(*sf)
    L.jxlsynth_set_features(pa, len(pf), sa, len(sf), num_extra)


SUBSAMPLING = {"444": (0, 0, 0), "420": (0, 1, 0), "422": (0, 2, 0), "440": (0, 3, 0), "mixed": (2, 1, 0)}


def encode_ycbcr(rgb, subsampling="420", seed=1, distance=1.0):
    """YCbCr VarDCT frame with chroma subsampling (tools/synth_ycbcr.h): rgb (h,w,3) uint8; subsampling a key of SUBSAMPLING or a
    (Cb, Y, Cr) tuple of sampling-factor modes (0 = 1x1, 1 = 2x2, 2 = 2x1, 3 = 1x2)."""
    L = lib()
    L.jxlsynth_ycbcr.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_uint32, C.c_float, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    h, w = rgb.shape[:2]
    a = np.ascontiguousarray(rgb, dtype=np.uint8)
    modes = (C.c_int32 * 3)(*(SUBSAMPLING[subsampling] if isinstance(subsampling, str) else subsampling))
    out = C.c_void_p(); n = C.c_size_t()
    if L.jxlsynth_ycbcr(a.ctypes.data, w, h, modes, seed, distance, C.byref(out), C.byref(n)):
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


def jpeg_transcode_codestream(w, h, modes, planes, qts):
    """Codestream of a lossless JPEG transcode (tools/synth_ycbcr.h EncodeJpegTranscode).  modes / planes / qts per jxl channel (Cb, Y, Cr):
    sampling-factor mode as in SUBSAMPLING, quantised coefficients int16 [blocks, 64] in JPEG natural order over the component's block
    grid (MCU grid x sampling factor), quantisation table int32 [64] natural order."""
    L = lib()
    L.jxlsynth_jpeg_transcode.argtypes = [C.c_int, C.c_int, C.POINTER(C.c_int32), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    m = (C.c_int32 * 3)(*modes)
    pl = [None if p is None else np.ascontiguousarray(p, dtype=np.int16) for p in planes]     # Cb = Cr = None: a grey JPEG
    q = np.ascontiguousarray(qts, dtype=np.int32)
    out = C.c_void_p(); n = C.c_size_t()
    ptr = [None if p is None else p.ctypes.data for p in pl]
    if L.jxlsynth_jpeg_transcode(w, h, m, ptr[0], ptr[1], ptr[2], q.ctypes.data, C.byref(out), C.byref(n)):
        raise RuntimeError(L.jxlsynth_last_error().decode())
    return _take(out, n)


def set_color(white_point=None, primaries=1, tf=13, gamma=0.0, intensity_target=255.0):
    """Enumerated colour encoding of the image headers written from now on (color_encoding_internal.h enums: white point 1 D65 / 10 E /
    11 DCI, primaries 1 sRGB / 9 BT.2100 / 11 P3, tf 1 Rec.709 / 8 linear / 13 sRGB / 16 PQ / 17 DCI / 18 HLG; gamma != 0 replaces tf)
    and the intensity target in nits; call without arguments to clear."""
    L = lib()
    L.jxlsynth_set_color.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_float]
    if white_point is None:
        L.jxlsynth_set_color(-1, 1, 13, 0, 255.0)
    else:
        L.jxlsynth_set_color(white_point, primaries, tf, int(round(gamma * 1e7)), intensity_target)


def set_animation(tps_num=0, tps_den=1, loops=0):
    """Image headers written from now on announce an animation (tps_num / tps_den ticks per second; 0: none); frames then carry frame(duration=...)."""
    lib().jxlsynth_set_animation(int(tps_num), int(tps_den), int(loops))


def set_preview(w=0, h=0):
    """The image headers written from now on announce a preview frame of w x h (0: none): the caller puts a frame of that size (emit=1) in front of the
    image's frames."""
    lib().jxlsynth_set_preview(int(w), int(h))


def set_lz77_lf(on=False):
    """VarDCT frames written from now on (this thread) code their LF-group Modular streams (LF coefficients, HF metadata) with LZ77: runs of equal
    values and repeats of the row above become copies."""
    lib().jxlsynth_set_lz77_lf(1 if on else 0)


def set_lz77_ac(on=False):
    """VarDCT frames written from now on (this thread) code their AC coefficient streams with LZ77 (runs of zero coefficients, repeating pairs and
    triples become copies; no special distances: the readers of these streams have no distance multiplier)"""
    lib().jxlsynth_set_lz77_ac(1 if on else 0)


def set_alpha_squeeze(on=False):
    """VarDCT frames written from now on (this thread) put their extra channel through the default Squeeze chain, like a cjxl encode of an RGBA picture with a progressive or lossy alpha: its
    sub-channels ride in GlobalModular, the LfGroup sections (between LF coefficients and HF metadata) and the PassGroup sections of the last pass"""
    lib().jxlsynth_set_alpha_squeeze(1 if on else 0)


def set_hf_presets(n=1):
    """VarDCT frames written from now on (this thread) carry n histogram sets for their AC coefficients (HfGlobal num_hf_presets); group g of every pass uses set g % n"""
    lib().jxlsynth_set_hf_presets(int(n))


def set_modular_group_shift(shift=1):
    """Modular frames written from now on (this thread) use groups of 128 << shift samples a side (frame header group_size_shift; 1 = 256 is the default)"""
    lib().jxlsynth_set_modular_group_shift(int(shift))


def set_custom_filters(on=False):
    """VarDCT frames written from now on (this thread) carry custom gaborish weights, EPF sharpness LUT, channel scales and sigma parameters in their RestorationFilter bundle"""
    lib().jxlsynth_set_custom_filters(1 if on else 0)


def set_custom_block_ctx(on=False):
    """VarDCT frames written from now on (this thread) carry their own BlockCtxMap: thresholds on the quantised LF of the three channels and on the quantiser field, 1404 entries onto 16 block contexts"""
    lib().jxlsynth_set_custom_block_ctx(1 if on else 0)


def set_custom_lf_global(on=False):
    """VarDCT frames written from now on (this thread) carry their own LF dequantisation steps and chroma-from-luma parameters (colour factor, base correlations, LF factors)"""
    lib().jxlsynth_set_custom_lf_global(1 if on else 0)


def set_custom_opsin(on=False):
    """XYB images written from now on (this thread) carry an OpsinInverseMatrix bundle of their own (matrix, opsin biases, quantisation biases as binary16 values)"""
    lib().jxlsynth_set_custom_opsin(1 if on else 0)


def set_qm_scales(x=3, b=2):
    """x_qm_scale / b_qm_scale (0..7) of the VarDCT frames written from now on (this thread): X / B quantisation steps times 0.8^(scale - 2)"""
    lib().jxlsynth_set_qm_scales(int(x), int(b))


def set_quant_lf(q=16):
    """quant_lf (1..65536) of the VarDCT frames written from now on (this thread)"""
    lib().jxlsynth_set_quant_lf(int(q))


def set_lf_extra_precision(e=0):
    """extra_precision (0..3) of the LF groups of the VarDCT frames written from now on (this thread): LF coefficients in steps 2^e times finer"""
    lib().jxlsynth_set_lf_extra_precision(int(e))


def set_lf_tree_shape(shape=0):
    """1: the LF-group streams of VarDCT frames written from now on (in this thread) use the MA-tree shape of a default-effort cjxl encode —
    weighted-predictor leaves under a fixed tree over property 15 for the LF coefficients, the fixed row / N / W tree for the HF metadata;
    0: the gradient tree (what `cjxl --faster_decoding` picks)."""
    lib().jxlsynth_set_lf_tree_shape(int(shape))


def set_prev_channel_props(on=False):
    """VarDCT frames written from now on (this thread): the MA tree of their LF-group streams also splits on previous-channel properties."""
    lib().jxlsynth_set_prev_channel_props(1 if on else 0)


def set_prefix(on=False):
    """Streams written from now on use prefix (Huffman) codes instead of ANS (cjxl -e 1..3); call without arguments to go back."""
    L = lib()
    L.jxlsynth_set_prefix.argtypes = [C.c_int]
    L.jxlsynth_set_prefix(1 if on else 0)


def set_float(exp_bits=0):
    """Float samples in the image headers written from now on: the Modular integers are bit patterns of floats with `exp_bits` exponent
    bits out of the `bits` the encoder is called with (0 = integer samples again)."""
    L = lib()
    L.jxlsynth_set_float.argtypes = [C.c_int]
    L.jxlsynth_set_float(int(exp_bits))


def set_spot(rgba=None):
    """The extra channel of the images written from now on is a spot colour (r, g, b, solidity — half-float precision) instead of alpha;
    call without arguments to go back to alpha."""
    L = lib()
    L.jxlsynth_set_spot.argtypes = [C.POINTER(C.c_float)]
    L.jxlsynth_set_spot(None if rgba is None else (C.c_float * 4)(*rgba))
