"""Round 6: stage times of a 128-frame batch (BASELINE config 3's per-GPU share) decoded alone, under the LF kernels: SIMT (stride 8), wave-wide (stride 64), SIMT + lf_wide_once."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
import jpegxl_rs_amd as jx
streams = bench.make_streams(32, 3840, 2160, 1)
for n in (128, 64, 32):
    for lf, wide, lpw in ((8, 0, 0), (64, 0, 0), (8, 1, 0), (64, 0, 4)):
        b = jx.BatchDecoder(0)
        b.add_many([streams[i % len(streams)] for i in range(n)], "uint8", 3, threads=8)
        b.set_lane_stride(lf, 1)
        b.prepare()
        if lpw: b.set_option("hf_lanes_per_wave", lpw)
        b.decode(); b.finish()
        b.collect_times()
        for _ in range(2):
            if wide: b.set_option("lf_wide_once", 1)
            b.decode_timed()
        b.finish()
        t, runs = b.collect_times()
        print(json.dumps({"frames": n, "lf_stride": lf, "wide_once": wide, "hf_lanes_per_wave": lpw, **{k: round(v / runs, 2) for k, v in t.items()}}), flush=True)
        del b
