#!/usr/bin/env python3
"""Runs phase A (gz_block_zeroing_orders) of one image, for rocprofv3 passes:
run_search.py W H [mask] [420]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
w, h = int(sys.argv[1]), int(sys.argv[2])
mask = int(sys.argv[3]) if len(sys.argv) > 3 else 7
L = guetzli_amd.load()
with L.context(images.tiled(w, h), 0.971769) as ctx:
    ctx.encode_rgb(download=False)
    if len(sys.argv) > 4:
        ctx.downsample(download=False)
    ctx.quantize(np.full((3, 64), 3, np.int32), download=False)
    ctx.block_zeroing_orders(comp_mask=mask)          # warm-up (mask of the original, tables)
    t0 = time.perf_counter()
    off, idx, err = ctx.block_zeroing_orders(comp_mask=mask)
    dt = time.perf_counter() - t0
    ev = ctx.search_evaluations()
    print(f"{w}x{h} mask {mask}: {len(idx)} candidates kept, {ev} CompareBlock evaluations in {dt*1e3:.2f} ms = {ev/dt/1e6:.1f} M evaluations/s")
