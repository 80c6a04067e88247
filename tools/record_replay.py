#!/usr/bin/env python3
"""Run one encode on the GPU through the record shim (tests/replay) and keep the log of
device results.  Usage: record_replay.py W H QUALITY OUT.log"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "replay"))
import build_replay, images
from guetzli_amd.encoder import HostLibrary
w, h, q, out = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
os.environ["GZ_REPLAY_MODE"] = "record"
os.environ["GZ_REPLAY_FILE"] = out
os.environ["GZ_REPLAY_REAL"] = os.path.join(ROOT, "guetzli_amd", "libguetzli_amd.so")
host = HostLibrary(build_replay.build_host())
rgb = images.bees() if (w, h) == (444, 258) else images.tiled(w, h)
t0 = time.perf_counter()
jpg, info = host.process(rgb, quality=q)
import json
sha = hashlib.sha256(jpg).hexdigest()
print(f"{w}x{h} q{q:g}: {len(jpg)} bytes sha256 {sha} in {time.perf_counter()-t0:.2f}s log {os.path.getsize(out)} bytes")
# sidecar for the CPU test that replays the log (tests/test_host_logic.py)
json.dump({"width": w, "height": h, "quality": q, "jpeg_bytes": len(jpg), "jpeg_sha256": sha,
           "iterations": info["counters"].get("number of iterations"),
           "head": open(os.path.join(ROOT, ".gpurun_head")).read().strip() if os.path.exists(os.path.join(ROOT, ".gpurun_head")) else None},
          open(out + ".json", "w"), indent=1)
