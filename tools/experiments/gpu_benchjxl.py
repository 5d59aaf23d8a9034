"""Times the reference's own benchmark input (benches/decode.rs:10, samples/bench.jxl: 2122x1433 Modular RGBA, weighted
predictor, per-group palettes) through the one-shot API and as a resident batch (not a pytest)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import torch
data = open(os.path.join(ROOT, "tests", "fixtures", "bench.jxl"), "rb").read()
dec = jx.decoder_builder()
dec.decode_with(data, np.uint8)
t = time.time()
for _ in range(5):
    meta, px = dec.decode_with(data, np.uint8)
one = (time.time() - t) / 5
b = jx.BatchDecoder(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for _ in range(n):
    b.add(data, dtype="uint8")
b.prepare()
for _ in range(2):
    b.decode()
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    b.decode()
torch.cuda.synchronize()
dev = (time.time() - t) / 3
npx = meta.width * meta.height
print("bench.jxl %dx%d: one-shot API %.1f ms (%.0f Mpx/s); resident batch of %d: %.1f ms (%.0f Mpx/s)" % (meta.width, meta.height, one * 1e3, npx / one / 1e6, n, dev * 1e3, n * npx / dev / 1e6))
