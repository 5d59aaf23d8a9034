"""Single-image decode latency on the GPU (not a pytest): BASELINE configs 2, 4, 5 through the one-shot JxlDecoder API and
through a resident one-image batch (device time only).  Prints one line per case."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import jpegxl_rs_amd as jx
import synth_lib as S
import torch


def smooth(seed, h, w, c, bits):
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    base = ((np.sin(xx / 37.0) + np.cos(yy / 23.0)) * 0.25 + 0.5) * ((1 << bits) - 1)
    return np.clip(base[..., None] + rng.normal(0, (1 << bits) / 1024.0, (h, w, c)).astype(np.float32), 0, (1 << bits) - 1).astype(np.int32)


cases = []
cases.append(("C2 3840x2160 VarDCT d1 u8", S.encode_vardct(S.synthetic_image(1000, 3840, 2160), seed=1000, strategy_mix=1, epf_iters=1, gab=1), np.uint8, 0, 3))
lin = ((S.synthetic_image(6, 7680, 4320).astype(np.float32) / 255.0) ** 2.2) * 4.0
cases.append(("C5 7680x4320 VarDCT HDR f32 epf3", S.encode_vardct(lin, seed=6, strategy_mix=1, epf_iters=3, gab=1, out_bits=32, hdr=1), np.float32, 2, 3))
cases.append(("C4 8192x8192 Modular squeeze u16 gray", S.encode_modular(smooth(6, 8192, 8192, 1, 16), 16, False, 1), np.uint16, 1, 1))
for name, data, dt, tcode, nch in cases:
    dec = jx.decoder_builder()
    dec.decode_with(data, dt)
    t = time.time(); meta, px = dec.decode_with(data, dt); one = time.time() - t
    b = jx.BatchDecoder(0)
    b.add(data, dtype=np.dtype(dt).name)
    b.prepare()
    for _ in range(2):
        b.decode(); torch.cuda.synchronize()
    t = time.time()
    for _ in range(5):
        b.decode()
    torch.cuda.synchronize()
    dev = (time.time() - t) / 5
    px_n = meta.width * meta.height
    print("%s: %d bytes; one-shot API %.1f ms (%.0f Mpx/s); resident decode %.2f ms (%.0f Mpx/s)" % (name, len(data), one * 1e3, px_n / one / 1e6, dev * 1e3, px_n / dev / 1e6), flush=True)
