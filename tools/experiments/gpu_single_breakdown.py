"""Where the single-image latency goes (not a pytest): one 3840x2160 VarDCT frame — one-shot API wall time, then the same frame as a one-image batch:
add (parse), prepare (tables, allocation, upload), decode stage by stage (HIP events), copy back."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import synth_lib as S
import torch

shape = int(os.environ.get("SHAPE", "0"))
if shape:
    S.set_lf_tree_shape(shape)
data = S.encode_vardct(S.synthetic_image(1000, 3840, 2160), seed=1000, strategy_mix=1, epf_iters=1, gab=1)
S.set_lf_tree_shape(0)
dec = jx.decoder_builder()
dec.decode_with(data, np.uint8)
ts = []
for _ in range(5):
    t = time.perf_counter(); dec.decode_with(data, np.uint8); ts.append((time.perf_counter() - t) * 1e3)
print("one-shot API ms:", [round(t, 1) for t in ts], flush=True)
for rep in range(3):
    t0 = time.perf_counter()
    b = jx.BatchDecoder(0)
    b.add(data, "uint8", 3)
    t1 = time.perf_counter()
    b.prepare()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    b.decode_timed(); b.finish()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    out = b.output(0)
    t4 = time.perf_counter()
    st, runs = b.collect_times()
    print("rep %d: create+add %.1f prepare %.1f decode %.1f (lf %.1f lfpost %.1f hf %.1f idct %.1f filter %.1f out %.1f) copy-back %.1f ms" % (
        rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, st["lf_ms"], st["lfpost_ms"], st["hf_ms"], st["idct_ms"], st["filter_ms"], st["out_ms"], (t4 - t3) * 1e3), flush=True)
    del b
