#!/usr/bin/env python3
"""SHA-256 of the UNMODIFIED reference's output for members of BASELINE config 5's batch (the
3840x2160 bench image circularly shifted by (37k rows, 53k cols), --quality 95): one process
per image, ~20 minutes of one core each (what `xargs -P` does in tests/golden_test.sh:24-26).
Usage: gen_config5_goldens.py K [K ...]   -> tests/golden/config5/k<K>.json"""
import hashlib, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import images
from checkers import ref
os.makedirs(os.path.join(ROOT, "tests", "golden", "config5"), exist_ok=True)
W, H = (int(os.environ.get("C5_W", 3840)), int(os.environ.get("C5_H", 2160)))
for k in (int(a) for a in sys.argv[1:]):
    rgb = images.shifted(images.tiled(W, H), k)
    t0 = time.time()
    jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(95.0), cap=W * H * 3 + (1 << 20))
    rec = {"k": k, "w": W, "h": H, "quality": 95.0, "rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
           "bytes": len(jpg), "jpeg_sha256": hashlib.sha256(jpg).hexdigest(), "reference_cpu_seconds": round(time.time() - t0, 1)}
    json.dump(rec, open(os.path.join(ROOT, "tests", "golden", "config5", f"k{k}.json"), "w"), indent=1)
    print(rec, flush=True)
