"""Run-time probe for a REAL libjxl on the box (SURVEY.md §8c last bullet) — TEST INFRASTRUCTURE ONLY.

The reference's arithmetic lives in libjxl v0.11.2, which is absent from /root/reference and from the build image.  If the machine
the tests / the bench run on happens to have one (system libjxl.so*, djxl/cjxl on PATH, a Python JXL plugin), it is the only
oracle that can pin the float pipeline: this module finds it and decodes files with it.  Everything runs in a SUBPROCESS — the
soname libjxl.so.0.11 collides with the look-alike this repo builds, and the two must never share a process.

    python tests/libjxl_probe.py                 # prints what was probed and what was found (JSON)
    python tests/libjxl_probe.py decode LIB FILE DTYPE NCH OUT.npy
"""
import ctypes as C
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OWN_LIB_DIR = os.path.realpath(os.path.join(ROOT, "jpegxl-rs_amd", "lib"))
LIB_DIRS = ["/usr/lib", "/usr/lib64", "/usr/local/lib", "/usr/local/lib64", "/lib", "/lib64", "/usr/lib/x86_64-linux-gnu", "/lib/x86_64-linux-gnu",
            "/opt/conda/lib", "/opt/rocm/lib", "/usr/lib/aarch64-linux-gnu"]
PY_MODULES = ["pillow_jxl", "jxlpy", "pyjxl", "imagecodecs", "pillow_jpegxl"]


def _candidate_libs():
    seen, out = set(), []
    dirs = list(LIB_DIRS) + [d for d in os.environ.get("LD_LIBRARY_PATH", "").split(":") if d]
    try:
        txt = subprocess.run(["ldconfig", "-p"], capture_output=True, text=True, timeout=20).stdout
        for line in txt.splitlines():
            if "libjxl.so" in line and "=>" in line:
                dirs.append(os.path.dirname(line.split("=>")[1].strip()))
    except Exception:
        pass
    for d in dirs:
        for p in sorted(glob.glob(os.path.join(d, "libjxl.so*"))):
            rp = os.path.realpath(p)
            if rp in seen or os.path.realpath(os.path.dirname(rp)) == OWN_LIB_DIR:
                continue
            seen.add(rp)
            out.append(p)
    return out, dirs


def _check_lib(path):
    """Loads `path` in a child process and asks for its version; a look-alike of this repo answers JxlHipLastError too."""
    code = ("import ctypes as C,sys\nL=C.CDLL(sys.argv[1])\nL.JxlDecoderVersion.restype=C.c_uint32\n"
            "print(L.JxlDecoderVersion(), int(hasattr(L,'JxlHipLastError')))\n")
    try:
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=60)
        v, own = r.stdout.split()
        return int(v), bool(int(own))
    except Exception:
        return None, False


def probe():
    """What exists on this machine.  Returns a dict with everything that was looked at (for the test log / the bench line)."""
    libs, dirs = _candidate_libs()
    found = {"probed": {"library_dirs": sorted(set(dirs)), "binaries": ["djxl", "cjxl"], "python_modules": PY_MODULES}, "lib": None, "lib_version": None,
             "djxl": shutil.which("djxl"), "cjxl": shutil.which("cjxl"), "python_module": None, "candidates_rejected": []}
    for p in libs:
        v, own = _check_lib(p)
        if v is None or own:
            found["candidates_rejected"].append({"path": p, "why": "look-alike of this repository" if own else "does not load"})
            continue
        found["lib"], found["lib_version"] = p, v
        break
    for m in PY_MODULES:
        code = "import importlib,sys\nm=importlib.import_module(sys.argv[1])\nprint(getattr(m,'__version__','?'))\n"
        if m == "imagecodecs":
            code = "import imagecodecs,sys\nassert imagecodecs.JPEGXL.available\nprint(imagecodecs.__version__)\n"
        try:
            r = subprocess.run([sys.executable, "-c", code, m], capture_output=True, text=True, timeout=120)
            if r.returncode == 0:
                found["python_module"] = m
                break
        except Exception:
            pass
    found["available"] = bool(found["lib"] or found["djxl"] or found["python_module"])
    return found


def describe(found):
    if found["available"]:
        return "real libjxl found: lib=%s (version %s) djxl=%s python=%s" % (found["lib"], found["lib_version"], found["djxl"], found["python_module"])
    p = found["probed"]
    return ("no libjxl on this box: probed libjxl.so* in %s (+ ldconfig -p), binaries %s on PATH, python modules %s%s" %
            (", ".join(p["library_dirs"]), "/".join(p["binaries"]), ", ".join(p["python_modules"]),
             "; rejected: " + ", ".join("%s (%s)" % (c["path"], c["why"]) for c in found["candidates_rejected"]) if found["candidates_rejected"] else ""))


# ---- decoding with the real library (child process) ---------------------------------------------------------------------------------
class _PixelFormat(C.Structure):
    _fields_ = [("num_channels", C.c_uint32), ("data_type", C.c_int), ("endianness", C.c_int), ("align", C.c_size_t)]


def _decode_with_lib(lib_path, data, dtype, nch):
    """The decode loop of jpegxl-rs/benches/decode.rs:16-37 against a real libjxl (one-shot, no runner)."""
    import numpy as np
    L = C.CDLL(lib_path)
    L.JxlDecoderCreate.restype = C.c_void_p
    L.JxlDecoderCreate.argtypes = [C.c_void_p]
    for n in ("JxlDecoderDestroy", "JxlDecoderCloseInput"):
        getattr(L, n).argtypes = [C.c_void_p]
    L.JxlDecoderSubscribeEvents.argtypes = [C.c_void_p, C.c_int]
    L.JxlDecoderSetInput.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    L.JxlDecoderProcessInput.argtypes = [C.c_void_p]
    L.JxlDecoderImageOutBufferSize.argtypes = [C.c_void_p, C.POINTER(_PixelFormat), C.POINTER(C.c_size_t)]
    L.JxlDecoderSetImageOutBuffer.argtypes = [C.c_void_p, C.POINTER(_PixelFormat), C.c_void_p, C.c_size_t]
    types = {"u8": (2, np.uint8), "u16": (3, np.uint16), "f32": (0, np.float32)}
    dt, npdt = types[dtype]
    dec = L.JxlDecoderCreate(None)
    try:
        assert L.JxlDecoderSubscribeEvents(dec, 0x1000) == 0
        buf = np.frombuffer(data, np.uint8)
        assert L.JxlDecoderSetInput(dec, buf.ctypes.data, len(data)) == 0
        L.JxlDecoderCloseInput(dec)
        fmt = _PixelFormat(nch, dt, 0, 0)
        out = None
        while True:
            st = L.JxlDecoderProcessInput(dec)
            if st == 5:
                size = C.c_size_t()
                assert L.JxlDecoderImageOutBufferSize(dec, C.byref(fmt), C.byref(size)) == 0
                out = np.zeros(size.value, np.uint8)
                assert L.JxlDecoderSetImageOutBuffer(dec, C.byref(fmt), out.ctypes.data, size.value) == 0
            elif st == 0x1000:
                continue
            elif st == 0:
                break
            else:
                raise RuntimeError("libjxl status %d" % st)
        return out.view(npdt)
    finally:
        L.JxlDecoderDestroy(dec)


def decode(found, data, dtype="u8", nch=3, timeout=600):
    """Decodes `data` with whatever real libjxl `found` names; returns a flat numpy array (interleaved, nch channels) or None."""
    import tempfile
    import numpy as np
    with tempfile.TemporaryDirectory() as tmp:
        src, dst = os.path.join(tmp, "in.jxl"), os.path.join(tmp, "out.npy")
        open(src, "wb").write(data)
        if found.get("lib"):
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "decode", found["lib"], src, dtype, str(nch), dst], capture_output=True, text=True, timeout=timeout)
            if r.returncode == 0 and os.path.exists(dst):
                return np.load(dst)
        if found.get("djxl") and dtype in ("u8", "u16") and nch in (1, 3):
            pnm = os.path.join(tmp, "out.ppm" if nch == 3 else "out.pgm")
            r = subprocess.run([found["djxl"], src, pnm, "--bits_per_sample", "8" if dtype == "u8" else "16"], capture_output=True, timeout=timeout)
            if r.returncode == 0 and os.path.exists(pnm):
                raw = open(pnm, "rb").read()
                parts = raw.split(None, 4) if nch == 1 else raw.split(None, 4)
                w, h, mx = int(parts[1]), int(parts[2]), int(parts[3])
                body = raw[len(raw) - w * h * nch * (2 if mx > 255 else 1):]
                return np.frombuffer(body, ">u2" if mx > 255 else np.uint8).astype(np.uint16 if mx > 255 else np.uint8)
    return None


if __name__ == "__main__":
    if len(sys.argv) >= 7 and sys.argv[1] == "decode":
        import numpy as np
        np.save(sys.argv[6], _decode_with_lib(sys.argv[2], open(sys.argv[3], "rb").read(), sys.argv[4], int(sys.argv[5])))
    else:
        f = probe()
        print(json.dumps(f, indent=1))
        print(describe(f))
