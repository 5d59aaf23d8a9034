"""Corruption loop over the stream kinds added at the end of round 2 (colour encodings, float samples, spot colours, float alpha): every
trial in-process; a crash shows up as a dead interpreter.  Not a pytest (run under gpurun)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_lib as S
import jpegxl_rs_amd as jx

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "11")))
img = S.synthetic_image(31, 200, 136)
rgba = rng.integers(0, 256, (72, 104, 4)).astype(np.int32)
streams = []
for kw in (dict(white_point=1, primaries=9, tf=16, intensity_target=1000.0), dict(white_point=1, primaries=9, tf=18, intensity_target=1000.0), dict(white_point=11, primaries=11, tf=17)):
    S.set_color(**kw); streams.append(S.encode_vardct(img, seed=5, strategy_mix=2)); S.set_color()
S.set_float(5); streams.append(S.encode_modular(rng.integers(0, 0x3C00, (60, 80, 4)).astype(np.int32), 16, False, 0)); S.set_float(0)
S.set_spot((1.0, 0.25, 0.125, 0.75)); streams.append(S.encode_modular(rgba, 8, False, 0)); streams.append(S.encode_vardct(S.synthetic_image(3, 104, 72), seed=5, alpha=rgba[..., 3].astype(np.uint8))); S.set_spot()
out = {"decoded": 0, "error": 0}
for data in streams:
    for trial in range(int(os.environ.get("FUZZ_TRIALS", "48"))):
        bad = bytearray(data)
        hi = len(bad) if trial % 2 else min(len(bad), 200)
        for pos in rng.integers(2, hi, 1 + trial % 3):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        for dtype in (np.uint8, np.float32):
            try:
                meta, px = jx.decoder_builder().decode_with(bytes(bad), dtype)
                out["decoded"] += 1
            except jx.DecodeError:
                out["error"] += 1
print(out, flush=True)
