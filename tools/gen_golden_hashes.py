#!/usr/bin/env python3
"""Generates tests/golden/jpeg_hashes.json: SHA-256 of the UNMODIFIED reference's output
(oracle/_ref, built from /root/reference) for whole-encode cases that are too slow to run
the reference on inside the test-suite.  Run where /root/reference exists (minutes of CPU)."""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import images
from checkers import ref
CASES = [("tiled", 1001, 777, 95.0), ("tiled", 612, 408, 88.0), ("synthetic", 640, 480, 95.0),
         ("synthetic", 333, 250, 84.0)]
out = {}
# JPEG input (guetzli::Process(params, stats, jpeg_data, &out)): the input stream is written by
# Pillow at run time; its hash is recorded so that a different libjpeg is noticed.
import io
from PIL import Image
for name, w, h, pil_kw, q, clear in [("jpegin_612x408_base", 612, 408, dict(quality=96, subsampling=0), 95.0, True),
                                     ("jpegin_500x333_prog_meta", 500, 333, dict(quality=97, subsampling=0, progressive=True, comment=b"golden"), 90.0, False)]:
    b = io.BytesIO()
    Image.fromarray(images.tiled(w, h)).save(b, "JPEG", **pil_kw)
    data = b.getvalue() + (b"" if clear else b"TAIL")
    t0 = time.time()
    jpg, _ = ref.process_jpeg(data, ref._butteraugli_score_for_quality(q), clear_metadata=clear)
    out[name] = {"kind": "jpeg", "w": w, "h": h, "pil": {k: (v.decode() if isinstance(v, bytes) else v) for k, v in pil_kw.items()},
                 "quality": q, "clear_metadata": clear, "input_sha256": hashlib.sha256(data).hexdigest(),
                 "bytes": len(jpg), "jpeg_sha256": hashlib.sha256(jpg).hexdigest()}
    print(name, out[name], f"{time.time() - t0:.0f}s", flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "jpeg_input_hashes.json"), "w"), indent=1)
sys.exit(0) if "--jpeg-only" in sys.argv else None
out = {}
for kind, w, h, q in CASES:
    rgb = images.tiled(w, h) if kind == "tiled" else images.synthetic(w, h)
    t0 = time.time()
    jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(q))
    key = f"{kind}_{w}x{h}_q{q:g}"
    out[key] = {"rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(), "bytes": len(jpg),
                "jpeg_sha256": hashlib.sha256(jpg).hexdigest()}
    print(key, out[key], f"{time.time() - t0:.0f}s", flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "jpeg_hashes.json"), "w"), indent=1)
