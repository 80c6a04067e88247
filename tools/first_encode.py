#!/usr/bin/env python3
"""Where the first encode of a process spends its time: timers of the first and of the second
1080p encode side by side (HIP start-up, code-object load, pool fill).  Usage: first_encode.py [W H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.perf_counter()
import guetzli_amd, images
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
rgb = images.tiled(w, h)
t1 = time.perf_counter()
L = guetzli_amd.load()
n = L.device_count()
t2 = time.perf_counter()
print(f"import + image {t1 - t0:.3f} s, load library + device_count {t2 - t1:.3f} s ({n} device)")
for i in range(3):
    t = time.perf_counter()
    jpg, info = guetzli_amd.process(rgb, quality=95)
    dt = time.perf_counter() - t
    tm = info["timers"]
    print(f"encode {i}: {dt:.3f} s; " + ", ".join(f"{k} {tm[k]:.3f}" for k in ("create+encode", "select_quant_matrix", "block_search", "compare", "phase_b_host", "jpeg_write", "total") if k in tm))
