"""Round 5: stage times (HIP events, nothing else on the GPU) of BASELINE config 5 frames (7680x4320, gaborish + EPF 3, f32 out).  N = frames (default 8).
A/B knobs of the library: JXL_HIP_NO_GAB_FOLD=1 (gaborish as a pass of its own instead of inside the first EPF pass)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
cache = os.environ.get("JXL_BENCH_STREAM_CACHE", "/tmp/sc"); os.makedirs(cache, exist_ok=True)
def get(seed):
    p = os.path.join(cache, f"hdr8k_{seed}.jxl")
    if os.path.exists(p): return open(p, "rb").read()
    d = bench._make_8k_hdr(seed); open(p, "wb").write(d); return d
streams = [get(s) for s in (6, 7, 8, 9)]
import jpegxl_rs_amd as jx
n = int(os.environ.get("N", "8"))
b = jx.BatchDecoder(0)
b.add_many([streams[i % len(streams)] for i in range(n)], "float32", 3, threads=8)
b.set_lane_stride(8, 1)
b.prepare()
b.decode(); b.finish()
b.decode_timed(); b.finish(); b.collect_times()
for _ in range(3):
    b.decode_timed(); b.finish()
t, runs = b.collect_times()
print(json.dumps({"frames": n, "no_gab_fold": os.environ.get("JXL_HIP_NO_GAB_FOLD"), **{k: round(v / max(runs, 1), 2) for k, v in t.items()}}))
