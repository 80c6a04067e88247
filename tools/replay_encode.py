#!/usr/bin/env python3
"""Replay an encode on the CPU from a log recorded on the GPU box (tests/replay): runs the
product's host search driver against the logged device results.  Host-logic profiling and
regression check.  Usage: replay_encode.py W H QUALITY LOG"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "replay"))
import build_replay, images
from guetzli_amd.encoder import HostLibrary
w, h, q, log = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
os.environ["GZ_REPLAY_MODE"] = "replay"
os.environ["GZ_REPLAY_FILE"] = log
host = HostLibrary(build_replay.build_host())
rgb = images.bees() if (w, h) == (444, 258) else images.tiled(w, h)
t0 = time.perf_counter()
jpg, info = host.process(rgb, quality=q)
dt = time.perf_counter() - t0
print(f"{w}x{h} q{q:g}: {len(jpg)} bytes sha256 {hashlib.sha256(jpg).hexdigest()} in {dt:.2f}s")
print("   timers:", {k: round(v, 3) for k, v in sorted(info["timers"].items(), key=lambda kv: -kv[1])})
