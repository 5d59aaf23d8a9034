"""Isolates crashing inputs of the JPEG-reconstruction corruption loop: every trial in a subprocess (not a pytest)."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if os.environ.get("FUZZ_CASE"):                    # a transcode of a libjpeg-written JPEG (tests/jpeg_cases.py) instead of the fixture
    import jpeg_cases as JC
    import jpeg_tools as J
    sel = os.environ["FUZZ_CASE"]
    if sel.startswith("p"):                        # progressive file
        data = J.transcode(JC.jpeg_bytes(JC.PROGRESSIVE[int(sel[1:])]))
    elif sel.startswith("g"):                      # grey, progressive
        data = J.transcode(JC.grey_jpeg_bytes(75, 52, 85, progressive=True))
    elif sel.startswith("m"):                      # ICC / Exif / XMP in the codestream / boxes (brob)
        import io
        from PIL import Image, ImageCms
        icc = ImageCms.ImageCmsProfile(ImageCms.createProfile("sRGB")).tobytes()
        ex = Image.Exif(); ex[0x010E] = "a test image"
        buf = io.BytesIO()
        Image.fromarray(JC.photo(67, 45)).save(buf, "JPEG", quality=85, subsampling=2, icc_profile=icc, exif=ex.tobytes(), xmp=b"<x:xmpmeta xmlns:x='adobe:ns:meta/'/>", progressive=True)
        data = J.transcode(buf.getvalue(), typed_metadata=True, compress_boxes=True)
    else:
        data = J.transcode(JC.jpeg_bytes(JC.CASES[int(sel)]))
else:
    data = open(os.path.join(ROOT, "tests", "fixtures", "sample_jpg.jxl"), "rb").read()


def mutate(t, seed):
    rng = np.random.default_rng(seed)
    for trial in range(t + 1):
        bad = bytearray(data)
        for pos in rng.integers(40, len(bad), 1 + trial % 2):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
    return bytes(bad)


if len(sys.argv) == 3:
    import jpegxl_rs_amd as jx
    try:
        meta, (kind, val) = jx.decoder_builder().reconstruct(mutate(int(sys.argv[1]), int(sys.argv[2])))
        print("ok", kind, len(val))
    except jx.DecodeError as e:
        print("error", str(e)[:100])
else:
    for seed in (321, 7):
        for t in range(60):
            r = subprocess.run([sys.executable, __file__, str(t), str(seed)], capture_output=True, text=True, timeout=120)
            last = (r.stdout.strip().splitlines() or [""])[-1]
            if not (last.startswith("ok") or last.startswith("error")):
                msg = [l for l in (r.stderr + r.stdout).splitlines() if "fault" in l.lower() or "abort" in l.lower() or "terminate" in l or "what()" in l or "Segmentation" in l]
                print("CRASH seed", seed, "trial", t, "rc", r.returncode, msg[:3], flush=True)
    print("done")
