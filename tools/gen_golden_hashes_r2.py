#!/usr/bin/env python3
"""Generates tests/golden/params_hashes.json: SHA-256 of the UNMODIFIED reference's output
(oracle/_ref, built from /root/reference) for whole encodes with non-default guetzli::Params
(YUV 4:2:0 modes, zeroing look-ahead, old zeroing model), greyscale input and YUV 4:2:0 JPEG
input -- cases that are too slow to run the reference on inside the GPU test-suite.  Run where
/root/reference exists (minutes of CPU)."""
import hashlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image
import images
from checkers import ref

def image(spec):
    kind, w, h = spec
    if kind == "bees":
        return images.bees()
    if kind == "grey":
        return np.repeat(images.tiled(w, h)[:, :, 1:2], 3, axis=2).copy()
    return images.tiled(w, h) if kind == "tiled" else images.synthetic(w, h)

CASES = [
    # name, image spec, quality, Params fields, Pillow kwargs for JPEG input (or None)
    ("bees_force420_q95", ("bees", 444, 258), 95.0, dict(force_420=True), None),
    ("bees_try420_q90", ("bees", 444, 258), 90.0, dict(try_420=True), None),
    ("tiled_333x251_force420_q84", ("tiled", 333, 251), 84.0, dict(force_420=True), None),
    ("synthetic_320x240_try420_q95", ("synthetic", 320, 240), 95.0, dict(try_420=True), None),
    ("grey_200x150_q95", ("grey", 200, 150), 95.0, dict(), None),
    ("grey_200x150_force420_q90", ("grey", 200, 150), 90.0, dict(force_420=True), None),
    ("tiled_300x200_lookahead2_q95", ("tiled", 300, 200), 95.0, dict(lookahead=2), None),
    ("tiled_300x200_lookahead5_oldmodel_q90", ("tiled", 300, 200), 90.0, dict(lookahead=5, new_model=False), None),
    ("tiled_300x200_lookahead1_q95", ("tiled", 300, 200), 95.0, dict(lookahead=1), None),
    ("jpegin420_612x408", ("tiled", 612, 408), 95.0, dict(), dict(quality=96, subsampling=2)),
    ("jpegin420_501x333_prog_meta", ("tiled", 501, 333), 90.0, dict(clear_metadata=False),
     dict(quality=97, subsampling=2, progressive=True, comment=b"golden")),
    ("jpegin444_try420_400x300", ("tiled", 400, 300), 95.0, dict(try_420=True), dict(quality=97, subsampling=0)),
    ("tiled_333x251_force420_silver_q90", ("tiled", 333, 251), 90.0, dict(force_420=True, silver=True), None),
    ("bees_try420_silver_q95", ("bees", 444, 258), 95.0, dict(try_420=True, silver=True), None),
]
path = os.path.join(ROOT, "tests", "golden", "params_hashes.json")
out = json.load(open(path)) if os.path.exists(path) and "--all" not in sys.argv else {}
for name, spec, q, params, pil_kw in CASES:
    if name in out:
        continue
    rgb = image(spec)
    h, w, _ = rgb.shape
    t0 = time.time()
    target = ref._butteraugli_score_for_quality(q)
    entry = {"image": list(spec), "quality": q, "params": params}
    if pil_kw is None:
        jpg, _ = ref.process_params(rgb, target, **params)
        entry["rgb_sha256"] = hashlib.sha256(rgb.tobytes()).hexdigest()
    else:
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", **pil_kw)
        data = b.getvalue() + (b"" if params.get("clear_metadata", True) else b"TAIL")
        jpg, _ = ref.process_params(data, target, **params)
        entry["pil"] = {k: (v.decode() if isinstance(v, bytes) else v) for k, v in pil_kw.items()}
        entry["input_sha256"] = hashlib.sha256(data).hexdigest()
    entry["bytes"] = len(jpg)
    entry["jpeg_sha256"] = hashlib.sha256(jpg).hexdigest()
    out[name] = entry
    print(name, entry, f"{time.time() - t0:.0f}s", flush=True)
    json.dump(out, open(path, "w"), indent=1)
