#!/usr/bin/env python3
"""Round 3: SHA-256 of the UNMODIFIED reference's output (oracle/_ref) for whole encodes of the
YUV 4:2:0 path and of another quality AT BASELINE SIZES (1920x1080, 3840x2160: the sizes at which
the paired row / column passes, the device-decided descent and the 4:2:0 chroma search run their
multi-chunk paths) -- minutes to tens of minutes of one CPU core each.  One file per case under
tests/golden/params_r3/ (the cases run in parallel processes); tests/test_gpu_parity.py reads
them beside params_hashes.json.   Usage: gen_golden_hashes_r3.py NAME [NAME ...]"""
import hashlib, io, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from PIL import Image
import images
from checkers import ref

CASES = {
    # name: (image spec, quality, Params fields, Pillow kwargs for JPEG input or None)
    "tiled_1920x1080_force420_q95": (("tiled", 1920, 1080), 95.0, dict(force_420=True), None),
    "tiled_1920x1080_try420_q90": (("tiled", 1920, 1080), 90.0, dict(try_420=True), None),
    "tiled_1920x1080_q84": (("tiled", 1920, 1080), 84.0, dict(), None),
    "tiled_1921x1083_q95": (("tiled", 1921, 1083), 95.0, dict(), None),
    "jpegin420_1920x1080": (("tiled", 1920, 1080), 95.0, dict(), dict(quality=96, subsampling=2)),
    "tiled_3840x2160_force420_q95": (("tiled", 3840, 2160), 95.0, dict(force_420=True), None),
}
out_dir = os.path.join(ROOT, "tests", "golden", "params_r3")
os.makedirs(out_dir, exist_ok=True)
for name in sys.argv[1:]:
    spec, q, params, pil_kw = CASES[name]
    rgb = images.tiled(spec[1], spec[2])
    t0 = time.time()
    target = ref._butteraugli_score_for_quality(q)
    entry = {"image": list(spec), "quality": q, "params": params}
    if pil_kw is None:
        jpg, _ = ref.process_params(rgb, target, **params)
        entry["rgb_sha256"] = hashlib.sha256(rgb.tobytes()).hexdigest()
    else:
        b = io.BytesIO()
        Image.fromarray(rgb).save(b, "JPEG", **pil_kw)
        data = b.getvalue()
        jpg, _ = ref.process_params(data, target, **params)
        entry["pil"] = dict(pil_kw)
        entry["input_sha256"] = hashlib.sha256(data).hexdigest()
    entry["bytes"] = len(jpg)
    entry["jpeg_sha256"] = hashlib.sha256(jpg).hexdigest()
    entry["reference_cpu_seconds"] = round(time.time() - t0, 1)
    json.dump(entry, open(os.path.join(out_dir, name + ".json"), "w"), indent=1)
    print(name, entry, flush=True)
