"""BASELINE config 5 (7680x4320 HDR, gaborish + EPF 3, f32 out) through the library pipeline: jobs of B frames, in_flight : lf_streams : hf_streams.
usage: gpu_hdr8k_jobs.py B:in_flight:lf_streams:hf_streams [...]"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, jpegxl_rs_amd as jx
cache = os.environ.get("JXL_BENCH_STREAM_CACHE", "/tmp/sc"); os.makedirs(cache, exist_ok=True)
def cached(name, fn, seed):
    p = os.path.join(cache, f"{name}_{seed}.jxl")
    if os.path.exists(p): return open(p, "rb").read()
    d = fn(seed); open(p, "wb").write(d); return d
streams = [cached("hdr8k", bench._make_8k_hdr, 6 + i) for i in range(4)]
W, H = 7680, 4320
for cfg in sys.argv[1:]:
    B, infl, lfs, hfs = map(int, cfg.split(":"))
    try:
        p = jx.Pipeline(0, timed=1, jobs_in_flight=infl, lf_streams=lfs, hf_streams=hfs, prepare_threads=3, parse_threads=8, reserve_frames=B, reserve_width=W, reserve_height=H, reserve_plane_sets=2)
        nout = 2          # (nobody reads the pixels here: two buffers, as bench.py at N = 1)
        outs = [torch.empty((B, H, W, 3), dtype=torch.float32, device="cuda:0") for _ in range(nout)]
        job = [streams[i % 4] for i in range(B)]
        def run(n):
            tickets = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                o = outs[k % nout]
                tickets.append(p.submit(job, "float32", 3, device_ptrs=[o[i].data_ptr() for i in range(B)]))
                if len(tickets) > infl: p.wait(tickets.pop(0))
            for t in tickets: p.wait(t)
            torch.cuda.synchronize(); return time.perf_counter() - t0
        run(1); run(p.info("slots")); p.collect_times()
        n = int(os.environ.get("JOBS", "12"))
        dt = run(n)
        t, runs = p.collect_times()
        print(json.dumps({"B": B, "in_flight": infl, "lf_streams": lfs, "hf_streams": hfs, "mpixel_per_s": round(B * W * H * n / dt / 1e6, 1), "ms_per_job": round(dt / n * 1e3, 1), "device_gb": round(p.info("device_bytes") / 2**30, 2),
                          "private_plane_jobs": p.info("private_plane_jobs"), "stage_ms": {k: round(v / max(runs, 1), 1) for k, v in t.items()}}), flush=True)
        p.close(); del p, outs; torch.cuda.empty_cache(); jx.arena_pool_trim()
    except Exception as ex:
        print(json.dumps({"cfg": cfg, "error": repr(ex)[:300]}), flush=True)
        torch.cuda.empty_cache(); jx.arena_pool_trim()
