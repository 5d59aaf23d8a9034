"""BASELINE config 4 (8192x8192 u16 lossless Modular, default Squeeze chain) through the library pipeline with jobs of B frames: Mpixel/s, device memory, stage times.
usage: gpu_mod8k_jobs.py B:in_flight [B:in_flight ...]"""
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
streams = [cached("mod8k", bench._make_8k_modular, 5 + i) for i in range(2)]
W = H = 8192
for cfg in sys.argv[1:]:
    B, infl = map(int, cfg.split(":"))
    try:
        p = jx.Pipeline(0, timed=1, jobs_in_flight=infl, lf_streams=max(1, infl), prepare_threads=3, parse_threads=8, reserve_frames=B, reserve_width=W, reserve_height=H)
        outs = [torch.empty((B, H, W), dtype=torch.int16, device="cuda:0") for _ in range(infl + 2)]
        job = [streams[i % 2] for i in range(B)]
        def run(n):
            tickets = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                o = outs[k % len(outs)]
                tickets.append(p.submit(job, "uint16", 1, device_ptrs=[o[i].data_ptr() for i in range(B)]))
                if len(tickets) > infl: p.wait(tickets.pop(0))
            for t in tickets: p.wait(t)
            torch.cuda.synchronize(); return time.perf_counter() - t0
        run(1); run(infl + 3); p.collect_times()
        n = max(4, 24 // B)
        dt = run(n)
        t, runs = p.collect_times()
        print(json.dumps({"B": B, "in_flight": infl, "mpixel_per_s": round(B * W * H * n / dt / 1e6, 1), "ms_per_job": round(dt / n * 1e3, 1), "device_gb": round(p.info("device_bytes") / 2**30, 2),
                          "stage_ms": {k: round(v / max(runs, 1), 1) for k, v in t.items()}}), flush=True)
        p.close(); del p, outs; torch.cuda.empty_cache(); jx.arena_pool_trim()
    except Exception as ex:
        print(json.dumps({"B": B, "in_flight": infl, "error": repr(ex)[:300]}), flush=True)
        torch.cuda.empty_cache(); jx.arena_pool_trim()
