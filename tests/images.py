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
