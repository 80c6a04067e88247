#!/usr/bin/env python3
"""Where does a cold process spend its time before the first kernel?  Times the first HIP
call, the first and second gz_create (+ destroy) at 4K and 1080p."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
t0 = time.perf_counter()
import numpy as np
import guetzli_amd, images
L = guetzli_amd.load()
t1 = time.perf_counter()
n = L.device_count()
t2 = time.perf_counter()
print(f"import+dlopen {t1-t0:.3f}s, first HIP call (device count) {t2-t1:.3f}s")
for (w, h) in ((3840, 2160), (3840, 2160), (1920, 1080), (1920, 1080)):
    rgb = images.tiled(w, h)
    a = time.perf_counter()
    ctx = L.context(rgb, 0.97)
    b = time.perf_counter()
    ctx.encode_rgb(download=False); ctx.synchronize()
    c = time.perf_counter()
    ctx.close()
    d = time.perf_counter()
    print(f"{w}x{h}: gz_create {b-a:.3f}s, encode_rgb {c-b:.3f}s, destroy {d-c:.3f}s")
