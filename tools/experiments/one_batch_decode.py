"""One prepared batch decoded a few times with nothing else on the GPU — what the PMC passes of a round profile (tools/scripts/profile_round5.sh).
usage: one_batch_decode.py <4k|jpeg|hdr8k|mod8k> <frames> <decodes>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
kind, n, reps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
cache = os.environ.get("JXL_BENCH_STREAM_CACHE", "/tmp/sc"); os.makedirs(cache, exist_ok=True)
def cached(name, fn, seed):
    p = os.path.join(cache, f"{name}_{seed}.jxl")
    if os.path.exists(p): return open(p, "rb").read()
    d = fn(seed); open(p, "wb").write(d); return d
if kind == "4k":
    streams, dtype, nch = bench.make_streams(min(n, 32), 3840, 2160, 1, tree_shape=int(os.environ.get("TREE_SHAPE", "0"))), "uint8", 3
elif kind == "jpeg":
    streams, dtype, nch = bench._pool_map(bench._make_ycbcr420, [700 + i for i in range(min(n, 8))]), "uint8", 3
elif kind == "hdr8k":
    streams, dtype, nch = [cached("hdr8k", bench._make_8k_hdr, 6 + i) for i in range(min(n, 4))], "float32", 3
else:
    streams, dtype, nch = [cached("mod8k", bench._make_8k_modular, 5 + i) for i in range(min(n, 2))], "uint16", 1
import jpegxl_rs_amd as jx
b = jx.BatchDecoder(0)
b.add_many([streams[i % len(streams)] for i in range(n)], dtype, nch, threads=8)
b.set_lane_stride(int(os.environ.get("LF_STRIDE", "8")), 1)
b.prepare()
if os.environ.get("HF_LPW"): b.set_option("hf_lanes_per_wave", int(os.environ["HF_LPW"]))      # (1: the wave-wide HF kernel — latency mode)
for _ in range(reps):
    b.decode(); b.finish()
print("decoded", n, "frames x", reps)
del b          # (a normal exit: the profiler writes its files in its exit hooks)
