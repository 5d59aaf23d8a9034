"""JPEG files for the transcode tests, written at run time by Pillow's libjpeg (an encoder this repository has no part in)."""
import io

import numpy as np

# (width, height, Pillow subsampling 0 = 4:4:4 / 1 = 4:2:2 / 2 = 4:2:0, quality, extra save() arguments)
CASES = [
    (40, 50, 0, 90, {}),
    (67, 45, 2, 85, {}),
    (64, 64, 1, 75, {}),
    (130, 77, 2, 95, dict(optimize=True)),
    (100, 60, 2, 80, dict(restart_marker_blocks=3)),
    (97, 33, 1, 60, dict(restart_marker_rows=1, comment=b"written by libjpeg")),
    (300, 280, 2, 88, {}),                      # 2 x 2 groups
    (2100, 24, 2, 70, {}),                      # two LF groups wide
]


def photo(w, h, seed=3):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 9.0 + y / 17.0), 128 + 90 * np.cos(x / 5.0 - y / 11.0), 60 + (x * 2 + y) % 190], -1)
    return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)


def jpeg_bytes(case):
    from PIL import Image
    w, h, ss, q, kw = case
    buf = io.BytesIO()
    Image.fromarray(photo(w, h, seed=w + h)).save(buf, "JPEG", quality=q, subsampling=ss, **kw)
    return buf.getvalue()


def pil_pixels(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)
