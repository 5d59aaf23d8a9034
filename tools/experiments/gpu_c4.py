"""Resident decode of BASELINE config 4 (8192x8192 Modular Squeeze u16) a few times — for rocprofv3 (not a pytest)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import synth_lib as S
import torch
rng = np.random.default_rng(6)
h = w = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
img = np.clip((((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 0.25 + 0.5) * 65535)[..., None] + rng.normal(0, 64, (h, w, 1)).astype(np.float32), 0, 65535).astype(np.int32)
data = S.encode_modular(img, 16, False, 1)
b = jx.BatchDecoder(0)
b.add(data, dtype="uint16")
b.prepare()
for _ in range(3):
    b.decode()
torch.cuda.synchronize()
t = time.time()
for _ in range(3):
    b.decode()
torch.cuda.synchronize()
print("resident decode %.1f ms" % ((time.time() - t) / 3 * 1e3))
