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

# progressive files (libjpeg's default scan script: spectral selection + successive approximation, 10 scans); "gratings" is built so
# that whole rows of blocks only carry correction bits in the refinement scans: libjpeg ends its end-of-band runs after 937 such bits,
# which the reconstruction data records as reset points
PROGRESSIVE = [
    (67, 45, 2, 85, dict(progressive=True)),
    (64, 64, 0, 75, dict(progressive=True)),
    (130, 77, 1, 95, dict(progressive=True, optimize=True)),
    (100, 60, 2, 80, dict(progressive=True, restart_marker_blocks=3)),
    (300, 280, 2, 88, dict(progressive=True)),
    (1024, 256, 0, 90, dict(progressive=True, image="gratings")),
]


def gratings(w, h):
    y, x = np.mgrid[0:h, 0:w]
    g = 128 + 60 * np.cos((2 * x + 1) * np.pi / 16) + 40 * np.cos((2 * x + 1) * 3 * np.pi / 16) + 30 * np.cos((2 * y + 1) * 2 * np.pi / 16)
    return np.clip(np.stack([g, g, g], -1), 0, 255).astype(np.uint8)


def photo(w, h, seed=3):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([128 + 100 * np.sin(x / 9.0 + y / 17.0), 128 + 90 * np.cos(x / 5.0 - y / 11.0), 60 + (x * 2 + y) % 190], -1)
    return np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)


def jpeg_bytes(case):
    from PIL import Image
    w, h, ss, q, kw = case
    kw = dict(kw)
    img = gratings(w, h) if kw.pop("image", None) == "gratings" else photo(w, h, seed=w + h)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, "JPEG", quality=q, subsampling=ss, **kw)
    return buf.getvalue()


def pil_pixels(data):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(data)).convert("RGB")).astype(int)


def grey_jpeg_bytes(w, h, quality, **kw):
    """One-component JPEG of the first channel of photo()."""
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(photo(w, h, seed=w + h)[:, :, 0]).save(buf, "JPEG", quality=quality, **kw)
    return buf.getvalue()
