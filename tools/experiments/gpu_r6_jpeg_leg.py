"""Round 6: the JPEG-transcode-shaped leg of bench.py alone (YCbCr 4:2:0 4K frames through the pipeline) — quick check of job size / depth."""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench, numpy as np, torch
import jpegxl_rs_amd as jx, oracle_lib as O
streams = bench._pool_map(bench._make_ycbcr420, [700 + i for i in range(8)])
W, H = 3840, 2160
for B, infl in (((256, 11),) if os.environ.get("JPEG_LEG_ONLY_FIRST") else ((64, 4), (128, 6), (256, 6), (256, 11))):
    try:
        p = jx.Pipeline(0, timed=1, jobs_in_flight=infl, lf_streams=infl, prepare_threads=3, parse_threads=8, reserve_frames=B, reserve_width=W, reserve_height=H)
        outs = [torch.empty((B, H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(2)]
        job = [streams[i % len(streams)] for i in range(B)]
        def run(n):
            tickets = []
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for k in range(n):
                o = outs[k % len(outs)]
                tickets.append(p.submit(job, "uint8", 3, device_ptrs=[o[i].data_ptr() for i in range(B)]))
                if len(tickets) > infl: p.wait(tickets.pop(0))
            for t in tickets: p.wait(t)
            torch.cuda.synchronize(); return time.perf_counter() - t0
        run(1); run(infl + 2); p.collect_times()
        n = 8
        dt = run(n)
        t, runs = p.collect_times()
        ref = O.decode(streams[1]).pixels("u8", 3)
        ok = bool(np.array_equal(outs[(n - 1) % len(outs)][1].cpu().numpy().reshape(ref.shape), ref))
        print(json.dumps({"B": B, "in_flight": infl, "mpixel_per_s": round(B * W * H * n / dt / 1e6, 1), "ms_per_job": round(dt / n * 1e3, 1), "device_gb": round(p.info("device_bytes") / 2**30, 2), "verified": ok,
                          "stage_ms": {k: round(v / max(runs, 1), 1) for k, v in t.items()}}), flush=True)
        p.close(); del p, outs; torch.cuda.empty_cache(); jx.arena_pool_trim()
    except Exception as ex:
        print(json.dumps({"B": B, "in_flight": infl, "error": repr(ex)[:300]}), flush=True)
        torch.cuda.empty_cache(); jx.arena_pool_trim()
