#!/usr/bin/env python3
"""Adds jpeg_sha256_reference to the sidecar of a replay log (tools/record_replay.py): the hash of
the JPEG the UNMODIFIED reference (oracle/_ref, built from /root/reference by oracle/Makefile)
produces for the same pixels.  CPU only, about 70 s for 960x540.
Usage: replay_reference_hash.py tests/golden/replay/encode_960x540_q95.json"""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import images
from checkers import ref
path = sys.argv[1]
meta = json.load(open(path))
rgb = images.tiled(meta["width"], meta["height"])
jpg = ref.process(rgb, ref._butteraugli_score_for_quality(float(meta["quality"])))
jpg = jpg[0] if isinstance(jpg, tuple) else jpg
meta["jpeg_sha256_reference"] = hashlib.sha256(jpg).hexdigest()
meta["jpeg_bytes_reference"] = len(jpg)
json.dump(meta, open(path, "w"), indent=1)
print(meta)
