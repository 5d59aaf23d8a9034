"""Stage times of one resident 3840x2160 decode with the two HF kernels (not a pytest): lane stride 64 = one group stream per
wavefront (default of the one-shot API: lowest latency), 1 = SIMT, one stream per lane (bench.py: highest throughput)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import synth_lib as S
import torch

data = S.encode_vardct(S.synthetic_image(1000, 3840, 2160), seed=1000, strategy_mix=1, epf_iters=1, gab=1)
for hf in (64, 1):
    b = jx.BatchDecoder(0)
    b.add(data, dtype="uint8")
    b.set_lane_stride(64, hf)
    b.prepare()
    for _ in range(2):
        b.decode(); torch.cuda.synchronize()
    for _ in range(3):
        b.decode_timed(None)
    torch.cuda.synchronize()
    b.finish()
    t, r = b.collect_times()
    print("lane_stride_hf", hf, {k: round(v / max(r, 1), 2) for k, v in t.items()})
