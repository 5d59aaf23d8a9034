"""Which corrupted stream of test_corrupted_round3_streams_fail_cleanly_or_decode brings the process down when the work arena is poisoned (JXL_HIP_POISON_WORK=1)?
Prints the stream / trial before each decode; the damaged stream of the last line is written to gpurun_out/poison_case.jxl."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_lib as S
import jpegxl_rs_amd as jx
from test_synth_roundtrip import preview_streams, lf_frame_streams, multipass_modular_streams
rng = np.random.default_rng(321)
S.set_prefix(True)
try:
    pfx = S.encode_vardct(S.synthetic_image(44, 320, 200), seed=5, strategy_mix=2, epf_iters=1, gab=1)
finally:
    S.set_prefix(False)
lf = lf_frame_streams()
streams = [preview_streams()[1][1], lf[0][1], lf[1][1], lf[3][1], multipass_modular_streams()[0][1], multipass_modular_streams()[5][1], pfx]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
for si, data in enumerate(streams):
    for trial in range(20):
        bad = bytearray(data)
        for pos in rng.integers(16, len(bad), 1 + trial % 4):
            bad[pos] ^= 1 << int(rng.integers(0, 8))
        if trial % 7 == 6:
            bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]
        open(os.path.join(ROOT, "gpurun_out", "poison_case.jxl"), "wb").write(bytes(bad))
        print("stream", si, "trial", trial, "bytes", len(bad), flush=True)
        try:
            jx.decoder_builder().decode_with(bytes(bad), np.uint8)
        except jx.DecodeError as e:
            print("   error:", str(e)[:100], flush=True)
print("no crash")
