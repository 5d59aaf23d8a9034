"""GPU helper (not a pytest file): SIMT LF decode — parity against the oracle on a few shapes, then the LF stage's time for a batch
of 4K frames at several lanes-per-wavefront settings next to the one-wavefront-per-stream kernel.
usage: python tools/experiments/gpu_lf_simt.py [frames] [distinct]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import jpegxl_rs_amd as jx
import oracle_lib as O
import synth_lib as S

ok = True
for (w, h, mix, seed) in [(320, 200, 2, 3), (64, 64, 0, 4), (600, 520, 1, 5), (2100, 2100, 1, 6), (24, 17, 0, 7), (8, 8, 0, 8), (2049, 16, 1, 9)]:
    data = S.encode_vardct(S.synthetic_image(seed, w, h), seed=seed, strategy_mix=mix, epf_iters=1, gab=1)
    ref = O.decode(data).pixels("u8", 3)
    for lf in (1, 4, 16):
        b = jx.BatchDecoder(0)
        for _ in range(3):
            b.add(data, "uint8", 3)
        b.set_lane_stride(lf, 1); b.prepare()
        simt = b.info_value("lf_simt_frames"), b.info_value("lf_simt_lanes"), b.info_value("lf_simt_waves")
        b.decode(); b.finish()
        same = all(np.array_equal(b.output(i), ref) for i in range(3))
        ok &= same and simt[0] == 3
        print(f"{w}x{h} mix {mix} lane stride {lf}: simt frames/lanes/waves {simt} -> {'ok' if same else 'MISMATCH'}", flush=True)
print("PARITY", "OK" if ok else "FAILED", flush=True)

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
distinct = int(sys.argv[2]) if len(sys.argv) > 2 else 8
sys.path.insert(0, ROOT)
import bench
streams = bench.make_streams(distinct, 3840, 2160, 1)
refs = {}
stream = torch.cuda.current_stream().cuda_stream
for lf in (64, 1, 2, 4, 8):
    b = jx.BatchDecoder(0)
    out = torch.empty((n, 2160, 3840, 3), dtype=torch.uint8, device="cuda")
    for i in range(n):
        b.add(streams[i % distinct], "uint8", 3, device_ptr=out.data_ptr() + i * 3840 * 2160 * 3)
    b.set_lane_stride(lf, 1); b.prepare(stream)
    b.decode(stream); b.finish(stream)
    for _ in range(2):
        b.decode_timed(stream)
    b.finish(stream)
    t, runs = b.collect_times()
    info = {k: b.info_value(k) for k in ("lf_simt_frames", "lf_simt_lanes", "lf_simt_waves")}
    got = out[0].cpu().numpy().reshape(-1)
    if 0 not in refs:
        refs[0] = O.decode(streams[0]).pixels("u8", 3)
    print(f"lane stride {lf}: {info} lf {t['lf_ms'] / runs:.2f} ms, lfpost {t['lfpost_ms'] / runs:.2f}, hf {t['hf_ms'] / runs:.2f}, idct {t['idct_ms'] / runs:.2f}, filter {t['filter_ms'] / runs:.2f}; "
          f"frame 0 {'ok' if np.array_equal(got, refs[0]) else 'MISMATCH'}", flush=True)
    del b, out
    torch.cuda.empty_cache()
