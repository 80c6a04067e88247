#!/usr/bin/env python3
"""End-to-end encode on the GPU: prints size, sha256 and the phase timers.
Usage: encode_time.py W H [quality] [force_420|try_420] [repeat]   (time = median of the runs after the first)"""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
w, h = int(sys.argv[1]), int(sys.argv[2])
q = float(sys.argv[3]) if len(sys.argv) > 3 else 95.0
rgb = images.bees() if (w, h) == (444, 258) else images.tiled(w, h)
kw = {a: True for a in sys.argv[4:] if a in ("force_420", "try_420")}
rep = max([int(a) for a in sys.argv[4:] if a.isdigit()] + [1])
times = []
for _ in range(rep):
    t0 = time.perf_counter()
    jpg, info = guetzli_amd.process(rgb, quality=q, **kw)
    times.append(time.perf_counter() - t0)
dt = sorted(times[1:] or times)[len(times[1:] or times) // 2]   # median without the first (warm-up) run
print(f"{w}x{h} q{q:g}: {len(jpg)} bytes sha256 {hashlib.sha256(jpg).hexdigest()} in {dt:.3f} s "
      f"= {w*h/1e6/dt:.3f} MPix/s; iters {info['counters']}")
print("   timers:", {k: round(v, 3) for k, v in sorted(info["timers"].items(), key=lambda kv: -kv[1])})
