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
for kind, w, h, q in CASES:
    rgb = images.tiled(w, h) if kind == "tiled" else images.synthetic(w, h)
    t0 = time.time()
    jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(q))
    key = f"{kind}_{w}x{h}_q{q:g}"
    out[key] = {"rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(), "bytes": len(jpg),
                "jpeg_sha256": hashlib.sha256(jpg).hexdigest()}
    print(key, out[key], f"{time.time() - t0:.0f}s", flush=True)
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "jpeg_hashes.json"), "w"), indent=1)
