"""Where does the host-side preparation of a fresh batch go (not a pytest): add (parse) vs prepare (allocation, upload, LF pre-run)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import jpegxl_rs_amd as jx
import synth_lib as S
W, H, n = 3840, 2160, int(sys.argv[1]) if len(sys.argv) > 1 else 128
streams = [S.encode_vardct(S.synthetic_image(1000 + i, W, H), seed=1000 + i, distance=1.0, epf_iters=1, gab=1, strategy_mix=1) for i in range(4)]
dst = torch.empty((n, H, W, 3), dtype=torch.uint8, device="cuda")
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    b = jx.BatchDecoder(0)
    t1 = time.perf_counter()
    for i in range(n):
        b.add(streams[i % 4], "uint8", 3, device_ptr=dst.data_ptr() + i * W * H * 3)
    t2 = time.perf_counter()
    b.prepare(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    b.decode(torch.cuda.current_stream().cuda_stream); b.finish(torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print("rep %d: create %.1f ms, add x%d %.1f ms, prepare %.1f ms, decode %.1f ms" % (rep, (t1 - t0) * 1e3, n, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
    del b
