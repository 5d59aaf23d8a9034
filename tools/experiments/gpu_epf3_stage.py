"""Round 5: filter-stage time of BASELINE config 5 frames (7680x4320, gaborish + EPF 3, f32 out): the tiled EPF kernels vs the per-pixel ones (JXL_HIP_EPF_STAGED=1)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import numpy as np
cache = os.environ.get("JXL_BENCH_STREAM_CACHE", "/tmp/sc"); os.makedirs(cache, exist_ok=True)
def get(seed):
    p = os.path.join(cache, f"hdr8k_{seed}.jxl")
    if os.path.exists(p): return open(p, "rb").read()
    d = bench._make_8k_hdr(seed); open(p, "wb").write(d); return d
streams = bench._pool_map(get, [6, 7, 8, 9])
import torch, jpegxl_rs_amd as jx
import oracle_lib as O
n = int(os.environ.get("N", "8"))
b = jx.BatchDecoder(0)
b.add_many([streams[i % len(streams)] for i in range(n)], "float32", 3, threads=8)
b.set_lane_stride(64, 1)
b.prepare(); b.decode(); b.finish(); b.collect_times()
for _ in range(3): b.decode_timed()
b.finish()
t, runs = b.collect_times()
print(json.dumps({"frames": n, "staged": os.environ.get("JXL_HIP_EPF_STAGED"), **{k: round(v / runs, 2) for k, v in t.items()}}))
if os.environ.get("VERIFY"):
    ref = O.decode(streams[0]).pixels("f32", 3)
    got = b.output(0)
    print("bit-exact vs oracle:", bool(np.array_equal(np.asarray(got).view(np.uint8).reshape(-1), np.asarray(ref).view(np.uint8).reshape(-1))))
