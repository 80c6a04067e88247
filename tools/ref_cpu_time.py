#!/usr/bin/env python3
"""The UNMODIFIED reference guetzli::Process (oracle/_ref/libgz_ref.so, 1 thread) timed on THIS
box's host CPU on a whole BASELINE image -- north_star: "the reference CPU path timed on the same
box's host cores (count stated)".  Far too long for bench.py (4.5 min at 1920x1080, ~20 min at
3840x2160), so it is run once per round through gpurun, pinned to one core, and its record kept
under profiles/ (bench.py quotes it in cpu_baseline.note).
Usage: ref_cpu_time.py W H [quality] > record.json"""
import hashlib, json, os, platform, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import images
from checkers import ref

w, h = int(sys.argv[1]), int(sys.argv[2])
q = float(sys.argv[3]) if len(sys.argv) > 3 else 95.0
model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
rgb = images.tiled(w, h)
t0 = time.perf_counter()
c0 = time.process_time()
jpg, _ = ref.process(rgb, ref._butteraugli_score_for_quality(q), cap=3 * w * h + (1 << 20))
dt, cpu = time.perf_counter() - t0, time.process_time() - c0
try:
    head = open(os.path.join(ROOT, ".gpurun_head")).read().strip()
except OSError:
    head = None
print(json.dumps({
    "workload": f"unmodified reference guetzli::Process, bees.png tiled to {w}x{h}, --quality {q:g}, 1 thread",
    "seconds": round(dt, 1), "cpu_seconds": round(cpu, 1), "value": round(w * h / 1e6 / dt, 6), "unit": "MPix/s",
    "cores_used": 1, "host_cpu": model, "host_cores_present": os.cpu_count(),
    "affinity": sorted(os.sched_getaffinity(0))[:8], "machine": platform.node(),
    "rgb_sha256": hashlib.sha256(rgb.tobytes()).hexdigest(),
    "output_bytes": len(jpg), "output_sha256": hashlib.sha256(jpg).hexdigest(),
    "note": "run beside the GPU test-suite of the same gpurun call (other cores); pinned with taskset",
    "head": head}, indent=1))
