"""Test / bench inputs.  All derived from tests/golden/bees.png (the reference's own
tests/bees.png, 444x258) exactly as SURVEY.md §8(d) prescribes: tiling from the origin
and cropping, circular shifts for the batch."""
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BEES = os.path.join(ROOT, "tests", "golden", "bees.png")

SHA_RGB = {
    (444, 258): "972b4d550d7d13038b8a1866cef112736f7826188bbdd5999ab8308d4eb471ae",
}


def bees():
    from PIL import Image
    im = np.array(Image.open(BEES).convert("RGB"))
    assert hashlib.sha256(im.tobytes()).hexdigest() == SHA_RGB[(444, 258)]
    return im


def tiled(w, h):
    """bees.png tiled from the origin, cropped to w x h (C2: 1920x1080, C3: 3840x2160)."""
    b = bees()
    bh, bw, _ = b.shape
    reps = (-(-h // bh), -(-w // bw), 1)
    return np.ascontiguousarray(np.tile(b, reps)[:h, :w])


def crop(w, h, x0=0, y0=0):
    return np.ascontiguousarray(bees()[y0:y0 + h, x0:x0 + w])


def shifted(img, k):
    """C5 batch member k: circular shift by (37k rows, 53k cols)."""
    return np.ascontiguousarray(np.roll(img, (37 * k, 53 * k), axis=(0, 1)))


def synthetic(w, h):
    """A deterministic non-photographic test image (no RNG): smooth gradients, a sinusoidal
    texture, hard-edged ellipses and a fine checker patch -- content unlike bees.png, to
    exercise the search on other statistics."""
    y, x = np.mgrid[0:h, 0:w].astype(np.float64)
    r = 127.5 + 127.5 * np.sin(x / 37.0) * np.cos(y / 53.0)
    g = 255.0 * (x / max(w - 1, 1)) * 0.6 + 60.0 * np.sin((x + 2 * y) / 11.0) ** 2
    b = 255.0 * (y / max(h - 1, 1))
    for cx, cy, rx, ry, col in ((0.3, 0.4, 0.18, 0.12, (230, 40, 60)), (0.7, 0.6, 0.1, 0.2, (20, 200, 120)),
                                (0.55, 0.25, 0.07, 0.07, (250, 250, 250))):
        m = ((x - cx * w) / (rx * w)) ** 2 + ((y - cy * h) / (ry * h)) ** 2 < 1.0
        r[m], g[m], b[m] = col
    ck = (x > 0.8 * w) & (y < 0.25 * h)
    v = 255.0 * (((x.astype(np.int64) // 2) + (y.astype(np.int64) // 2)) % 2)
    r[ck], g[ck], b[ck] = v[ck], v[ck], 255.0 - v[ck]
    return np.ascontiguousarray(np.clip(np.stack([r, g, b], -1), 0, 255).astype(np.uint8))


# ---- degenerate content (round 4): what the reference handles at the edges of its search ----
def flat(w, h, rgb):
    """One colour everywhere: all AC coefficients zero, nothing for the zeroing search to rank."""
    return np.ascontiguousarray(np.broadcast_to(np.array(rgb, np.uint8), (h, w, 3)))


def stripes(w, h, period=37):
    """Vertical stripes of the eight saturated corner colours of the RGB cube, `period` pixels
    wide (not a multiple of 8: every stripe edge falls inside a block)."""
    cols = np.array([(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (0, 255, 255),
                     (255, 0, 255), (0, 0, 0), (255, 255, 255)], np.uint8)
    idx = (np.arange(w) // period) % 8
    return np.ascontiguousarray(np.broadcast_to(cols[idx][None, :, :], (h, w, 3)))


def noise(w, h, seed=1):
    """Uniform noise without an RNG library: a 32-bit integer hash of the sample index
    (multiply / xorshift rounds), its top byte.  Every block has 63 non-zero AC coefficients per
    component -> the 189-candidate lists of ComputeBlockZeroingOrder."""
    i = (np.arange(w * h * 3, dtype=np.uint64) + np.uint64(seed) * np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF)
    for mul in (0x7FEB352D, 0x846CA68B):
        i ^= i >> np.uint64(16)
        i = (i * np.uint64(mul)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(16)
    return np.ascontiguousarray((i >> np.uint64(24)).astype(np.uint8).reshape(h, w, 3))


# ---- photographs (round 5): tests/golden/photos/, licences in LICENSES.md there ----------------
PHOTOS = os.path.join(ROOT, "tests", "golden", "photos")
PHOTO_FILES = {"china": "china.jpg", "flower": "flower.jpg", "astronaut": "astronaut.png",
               "coffee": "coffee.png", "chelsea": "chelsea.png", "gravel": "gravel.png",
               "rocket": "rocket.jpg", "hubble": "hubble_deep_field.jpg", "retina": "retina.jpg"}


def photo_bytes(name):
    return open(os.path.join(PHOTOS, PHOTO_FILES[name]), "rb").read()


def photo(name):
    """uint8 [h][w][3] of a committed photograph (a JPEG file decoded by Pillow's libjpeg: the
    fixtures record the SHA-256 of these pixels, a test skips on another decoder's rounding)."""
    from PIL import Image
    return np.ascontiguousarray(np.array(Image.open(os.path.join(PHOTOS, PHOTO_FILES[name])).convert("RGB")))


def mosaic(w, h):
    """A w x h image without any period: the nine photographs laid out in strips, left to right
    and top to bottom, each strip as high as its first photograph, the sequence continuing where
    the strip before stopped and a photograph cut where it meets the right or lower edge; every
    other strip starts with a horizontally mirrored copy shifted by a third of its width.  No RNG.
    (tests/images.tiled repeats a 444x258 tile 8.6 x 8.4 times at 3840x2160: this is the image
    that shows whether the 4K numbers hold off periodic content.)"""
    names = ["astronaut", "chelsea", "china", "coffee", "hubble", "gravel", "flower", "retina", "rocket"]
    pics = [photo(n) for n in names]
    out = np.zeros((h, w, 3), np.uint8)
    k, y, strip = 0, 0, 0
    while y < h:
        sh = min(pics[k % len(pics)].shape[0], h - y)
        x = 0
        first = True
        while x < w:
            p = pics[k % len(pics)]
            if strip % 2 == 1:
                p = p[:, ::-1]
            if first and strip % 2 == 1:
                p = p[:, p.shape[1] // 3:]
            first = False
            ph = min(p.shape[0], sh)
            pw = min(p.shape[1], w - x)
            out[y:y + ph, x:x + pw] = p[:ph, :pw]
            if ph < sh:      # a shorter photograph: continue it with its own mirrored rows
                rest = sh - ph
                fill = p[::-1][:rest, :pw]
                out[y + ph:y + ph + fill.shape[0], x:x + pw] = fill
                if fill.shape[0] < rest:
                    out[y + ph + fill.shape[0]:y + sh, x:x + pw] = p[:rest - fill.shape[0], :pw]
            x += pw
            k += 1
        y += sh
        strip += 1
        k += 2               # the next strip starts two photographs further on
    return np.ascontiguousarray(out)
