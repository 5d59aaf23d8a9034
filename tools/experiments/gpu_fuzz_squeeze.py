"""Corruption loop over squeezed extra channels in VarDCT frames (GlobalModular / LfGroup / PassGroup residual streams + the inverse Squeeze), alone and in batches of 12
beside sound frames.  Every trial in-process; a crash shows up as a dead interpreter.  Not a pytest (run under gpurun, inside `timeout`; JXL_HIP_POISON_WORK=1 fills
the work arena with 0xA5 first)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import oracle_lib as O
from test_synth_roundtrip import squeezed_alpha_streams

rng = np.random.default_rng(int(os.environ.get("FUZZ_SEED", "29")))
cases = squeezed_alpha_streams()
trials = int(os.environ.get("FUZZ_TRIALS", "80"))
out = {"decoded": 0, "error": 0}


def damage(data, trial):
    bad = bytearray(data)
    lo = 2 if trial % 3 == 0 else len(bad) // 5
    for pos in rng.integers(lo, len(bad), 1 + trial % 4):
        bad[pos] ^= 1 << int(rng.integers(0, 8))
    if trial % 9 == 8:
        bad = bad[: int(rng.integers(len(bad) // 3, len(bad)))]
    return bytes(bad)


for name, sq, plain, al in cases:
    for trial in range(trials):
        try:
            jx.decoder_builder().decode_with(damage(sq, trial), np.uint8)
            out["decoded"] += 1
        except jx.DecodeError:
            out["error"] += 1
    print(name, out, flush=True)
for name, sq, plain, al in (cases[1], cases[2], cases[5]):
    ref = O.decode(sq).pixels("u8", 4)
    for rep in range(int(os.environ.get("FUZZ_BATCHES", "10"))):
        bad_at = set(int(v) for v in rng.integers(0, 12, 3))
        b = jx.BatchDecoder(0)
        ok_add = []
        for i in range(12):
            try:
                b.add(damage(sq, rep * 12 + i) if i in bad_at else sq, "uint8", 4)
                ok_add.append(i)
            except jx.DecodeError:
                pass
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
for name, sq, plain, al in cases:
    meta, px = jx.decoder_builder().decode_with(sq, np.uint8)
    assert np.array_equal(px.reshape(-1), O.decode(sq).pixels("u8", 4)), "decoder unhealthy after the fuzz loop"
print("done", out, flush=True)
