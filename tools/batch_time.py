#!/usr/bin/env python3
"""Several images in flight on one GPU, timed in a warm process: batch_time.py W H N [workers] [repeats]
prints the MPix/s of every repetition after the first (A/B of runtime knobs such as
GPU_MAX_HW_QUEUES, which is read when the HIP runtime starts)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import guetzli_amd, images
from guetzli_amd.batch import encode_concurrent
w, h, n = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
workers = int(sys.argv[4]) if len(sys.argv) > 4 else 8
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
base = images.tiled(w, h)
imgs = [np.ascontiguousarray(np.roll(base, (37 * k, 53 * k), axis=(0, 1))) for k in range(n)]
rates = []
for r in range(reps + 1):
    t0 = time.perf_counter()
    out = encode_concurrent(imgs, lambda rgb: guetzli_amd.process(rgb, quality=95.0), workers=workers)
    dt = time.perf_counter() - t0
    if r: rates.append(n * w * h / 1e6 / dt)
print(f"{n} x {w}x{h}, {workers} in flight, GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')}: "
      + " ".join(f"{x:.2f}" for x in rates) + " MPix/s")
