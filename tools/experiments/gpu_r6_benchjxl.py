"""Round 6: bench.jxl (the reference's criterion input) through decode_with, a few times — for rocprofv3 --kernel-trace --stats."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import jpegxl_rs_amd as jx
data = open(os.path.join(ROOT, "tests", "fixtures", "bench.jxl"), "rb").read()
d = jx.decoder_builder()
for i in range(3):
    t0 = time.perf_counter(); meta, px = d.decode_with(data, np.uint8); t1 = time.perf_counter()
    print(json.dumps({"bench_jxl_ms": round((t1 - t0) * 1e3, 1), "w": meta.width, "h": meta.height}), flush=True)
