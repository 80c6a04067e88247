#!/usr/bin/env python3
"""Runs ITERS butteraugli Compare chains of one candidate (for rocprofv3 kernel-trace / PMC
passes): run_compare.py W H ITERS"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
w, h, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
L = guetzli_amd.load()
with L.context(images.tiled(w, h), 0.971769) as ctx:
    ctx.encode_rgb(download=False)
    ctx.quantize(np.full((3, 64), 3, np.int32), download=False)
    ms = ctx.time_compare(iters) / iters
    print(f"{w}x{h}: {ms:.4f} ms per Compare, {494.0 * w * h / ms / 1e6:.1f} GB/s algorithmic, distance {ctx.last_distance():.6f}")
