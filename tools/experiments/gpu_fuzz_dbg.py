"""Isolates crashing inputs of the corruption tests: every (stream, trial) in a subprocess (not a pytest).
usage: gpu_fuzz_dbg.py            -> runs all, prints the ones that do not end in 'decoded' / 'error'
       gpu_fuzz_dbg.py <i> <t>    -> runs one and prints the outcome"""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))


def streams():
    import pytest  # noqa: F401  (test module imports it)
    import synth_lib as S
    import test_gpu_parity as T
    from conftest import fixture_bytes
    from free_cases import FREE_CASES
    feats = T._feature_streams()
    img = S.synthetic_image(70, 300, 280)
    out = [fixture_bytes("sample_grey.jxl"), fixture_bytes("2bit.jxl"), feats["patches_splines_noise"], feats["patches_alpha_modes"], feats["patches_modular"],
           fixture_bytes("sample.jxl"), fixture_bytes("sample_jpg.jxl"), S.encode_vardct(img, seed=2, strategy_mix=2, epf_iters=2, gab=1, num_passes=2), S.encode_vardct(img, seed=2, strategy_mix=4, upsampling=2),
           S.encode_ycbcr(img, "420", seed=3), S.encode_ycbcr(S.synthetic_image(71, 203, 139), "mixed", seed=4)]
    for name in ("gray_alpha_16bit_everything", "lz77_local_trees", "palette_delta_wp_sections", "previous_channel_properties_groups", "local_tree_everywhere"):
        out.append(S.encode_modular_free(**dict(FREE_CASES[name], bits=16)))
    return out


SEED = int(os.environ.get("FUZZ_SEED", "321"))
TRIALS = int(os.environ.get("FUZZ_TRIALS", "16"))
FIRST = int(os.environ.get("FUZZ_FIRST_BYTE", "12"))       # 2: headers are mutated as well (only the signature is kept)


def mutate(all_streams, i, t):
    rng = np.random.default_rng(SEED)
    for si, data in enumerate(all_streams):
        for trial in range(TRIALS):
            bad = bytearray(data)
            hi = len(bad) if trial % 2 else min(len(bad), 400)          # every other trial hits the headers / first section
            for pos in rng.integers(FIRST, hi, 1 + trial % 3):
                bad[pos] ^= 1 << int(rng.integers(0, 8))
            if trial % 8 == 7:
                bad = bad[: int(rng.integers(len(bad) // 2, len(bad)))]
            if si == i and trial == t:
                return bytes(bad)


if len(sys.argv) == 3:
    import jpegxl_rs_amd as jx
    data = mutate(streams(), int(sys.argv[1]), int(sys.argv[2]))
    try:
        meta, px = jx.decoder_builder().decode_with(data, np.uint8)
        print("decoded", len(px))
    except jx.DecodeError as e:
        print("error", str(e)[:100])
else:
    n = len(streams())
    for i in range(n):
        for t in range(TRIALS):
            r = subprocess.run([sys.executable, __file__, str(i), str(t)], capture_output=True, text=True, timeout=120)
            last = (r.stdout.strip().splitlines() or [""])[-1]
            if not (last.startswith("decoded") or last.startswith("error")):
                msg = [l for l in (r.stderr + r.stdout).splitlines() if "fault" in l.lower() or "Error" in l or "abort" in l.lower() or "terminate" in l or "what()" in l]
                print("CRASH stream", i, "trial", t, "rc", r.returncode, msg[:3])
    print("done")
