"""round 6: single-frame latency (decode_with) before and after each of the big legs of bench.py in the same process — which one leaves the process slower?"""
import os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import bench, jpegxl_rs_amd as jx
W, H = 3840, 2160
streams = bench.make_streams(4, W, H, 1)
dec = jx.decoder_builder()
def lat(tag):
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); dec.decode_with(streams[0], np.uint8); ts.append((time.perf_counter() - t0) * 1e3)
    print(json.dumps({"after": tag, "single_frame_ms": round(sorted(ts[1:])[len(ts[1:]) // 2], 1), "pool_held_gb": round(jx.libjxl().JxlHipArenaPoolHeld() / 2**30, 1),
                      "torch_reserved_gb": round(torch.cuda.memory_reserved() / 2**30, 1), "free_gb": round(torch.cuda.mem_get_info()[0] / 2**30, 1)}), flush=True)
def run_leg(job, dtype, nch, B, infl, shape, tdt, n=4, **kw):
    p = jx.Pipeline(0, jobs_in_flight=infl, lf_streams=infl, prepare_threads=3, parse_threads=8, reserve_frames=B, reserve_width=shape[1], reserve_height=shape[0], **kw)
    outs = [torch.empty((B,) + shape, dtype=tdt, device="cuda:0") for _ in range(2)]
    tickets = []
    for k in range(n):
        o = outs[k % 2]
        tickets.append(p.submit(job, dtype, nch, device_ptrs=[o[i].data_ptr() for i in range(B)]))
        if len(tickets) > infl: p.wait(tickets.pop(0))
    for t in tickets: p.wait(t)
    p.close(); del p, outs
    torch.cuda.empty_cache(); jx.arena_pool_trim()
import ctypes
jx.libjxl().JxlHipArenaPoolHeld.restype = ctypes.c_size_t
lat("start")
which = sys.argv[1:] or ["mod8k", "jpeg", "hdr8k", "4k"]
for w in which:
    if w == "mod8k":
        s = [bench._make_8k_modular(5), bench._make_8k_modular(6)]
        run_leg([s[i % 2] for i in range(8)], "uint16", 1, 8, 2, (8192, 8192), torch.int16)
    elif w == "jpeg":
        s = bench._pool_map(bench._make_ycbcr420, [700 + i for i in range(4)])
        run_leg([s[i % 4] for i in range(128)], "uint8", 3, 128, 6, (H, W, 3), torch.uint8)
    elif w == "hdr8k":
        s = [bench._make_8k_hdr(6), bench._make_8k_hdr(7)]
        run_leg([s[i % 2] for i in range(32)], "float32", 3, 32, 6, (4320, 7680, 3), torch.float32, reserve_plane_sets=2)
    elif w == "4k":
        run_leg([streams[i % 4] for i in range(128)], "uint8", 3, 128, 11, (H, W, 3), torch.uint8, n=14)
    lat(w)
