"""Corruption loop over the stream kinds round 3 added (preview frame, LF frames of both encodings and two levels, Modular passes, prefix-coded progressive /
subsampled frames, LZ77-coded LF streams, previous-channel properties, animations): every trial in-process; a crash shows up as a dead interpreter.
Not a pytest (run under gpurun, inside `timeout`)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_lib as S
import jpegxl_rs_amd as jx
from test_synth_roundtrip import preview_streams, lf_frame_streams, multipass_modular_streams, lz77_lf_streams, prev_channel_streams

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "17")))
img = S.synthetic_image(41, 520, 300)
streams = [preview_streams()[0][1], preview_streams()[1][1]] + [c[1] for c in lf_frame_streams()[:3]] + [multipass_modular_streams()[k][1] for k in (0, 2, 4, 5)]
streams += [lz77_lf_streams()[0][1], lz77_lf_streams()[2][1], prev_channel_streams()[0][1]]
S.set_prefix(True)
try:
    streams += [S.encode_vardct(img, seed=5, strategy_mix=2, num_passes=3), S.encode_ycbcr(img, subsampling="420", seed=3)]
finally:
    S.set_prefix(False)
S.set_animation(100, 1, 0)
try:
    streams.append(S.encode_vardct_frame(S.synthetic_image(6, 300, 200), S.frame(is_last=0, save_as_reference=1, duration=10), seed=3)
                   + S.encode_vardct_frame(S.synthetic_image(9, 64, 48), S.frame(emit=1, have_crop=1, crop_x0=100, crop_y0=60, canvas_w=300, canvas_h=200, blend_mode=1, blend_source=1, duration=5), seed=4))
finally:
    S.set_animation(0)
out = {"decoded": 0, "error": 0}
trials = int(os.environ.get("FUZZ_TRIALS", "60"))
for si, data in enumerate(streams):
    for trial in range(trials):
        bad = bytearray(data)
        hi = len(bad) if trial % 2 else min(len(bad), 400)
        for pos in rng.integers(2, hi, 1 + trial % 3):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        if trial % 11 == 10:
            bad = bad[: int(rng.integers(len(bad) // 3, len(bad)))]
        try:
            meta, px = jx.decoder_builder().decode_with(bytes(bad), np.uint8)
            out["decoded"] += 1
        except jx.DecodeError:
            out["error"] += 1
    print(si, out, flush=True)
# the decoder is still healthy
import oracle_lib as O
for data in streams[:6]:
    meta, px = jx.decoder_builder().decode_with(data, np.uint8)
    assert np.array_equal(px.reshape(-1), O.decode(data).pixels("u8", 4 if meta.has_alpha_channel else 3)), "decoder unhealthy after the fuzz loop"
print("done", out, flush=True)
