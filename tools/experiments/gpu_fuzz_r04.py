"""Corruption loop over what round 4 added: LZ77-coded AC streams (single pass, progressive, prefix codes underneath, with extra channels), cjxl-shaped weighted-predictor LF
trees on the quad SIMT kernel (alone and in batches of 16 so that lanes of one wavefront fail independently), containers walked through the box API.  Every trial
in-process; a crash shows up as a dead interpreter.  Not a pytest (run under gpurun, inside `timeout`)."""
import ctypes as C
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth_lib as S
import jpegxl_rs_amd as jx
import oracle_lib as O
from test_synth_roundtrip import lz77_ac_streams, cjxl_shape_streams

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "23")))
lz = lz77_ac_streams()
cj = cjxl_shape_streams()
streams = [c[1] for c in lz] + [cj[0][2], cj[1][2], cj[4][2], cj[6][2]]
out = {"decoded": 0, "error": 0}
trials = int(os.environ.get("FUZZ_TRIALS", "60"))


def damage(data, trial):
    bad = bytearray(data)
    lo = 2 if trial % 3 == 0 else len(bad) // 4          # a third of the trials hit the headers too
    for pos in rng.integers(lo, len(bad), 1 + trial % 4):
        bad[pos] ^= 1 << int(rng.integers(0, 8))
    if trial % 11 == 10:
        bad = bad[: int(rng.integers(len(bad) // 3, len(bad)))]
    return bytes(bad)


for si, data in enumerate(streams):
    for trial in range(trials):
        try:
            meta, px = jx.decoder_builder().decode_with(damage(data, trial), np.uint8)
            out["decoded"] += 1
        except jx.DecodeError:
            out["error"] += 1
    print(si, out, flush=True)
# batches: sixteen copies of a weighted-predictor frame / an LZ77 frame, a few of them damaged — the others must come out right
for name, data in (("wp", cj[4][2]), ("lz77", lz[2][1])):
    ref = O.decode(data).pixels("u8", 3)
    for rep in range(int(os.environ.get("FUZZ_BATCHES", "12"))):
        bad_at = set(int(v) for v in rng.integers(0, 16, 3))
        b = jx.BatchDecoder(0)
        ok_add = []
        for i in range(16):
            try:
                b.add(damage(data, rep * 16 + i) if i in bad_at else data, "uint8", 3)
                ok_add.append(i)
            except jx.DecodeError:
                pass
        b.set_lane_stride(8, 1)
        try:
            b.prepare(); b.decode()
            try:
                b.finish()
            except jx.DecodeError:
                pass
        except jx.DecodeError:
            continue
        for k, i in enumerate(ok_add):
            if i not in bad_at:
                assert np.array_equal(b.output(k), ref), (name, rep, i)
    print("batches", name, "ok", flush=True)
for data in streams[:4]:
    meta, px = jx.decoder_builder().decode_with(data, np.uint8)
    assert np.array_equal(px.reshape(-1), O.decode(data).pixels("u8", 4 if meta.has_alpha_channel else 3)), "decoder unhealthy after the fuzz loop"
print("done", out, flush=True)
