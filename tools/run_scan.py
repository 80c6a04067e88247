#!/usr/bin/env python3
"""Times the device entropy coder alone (gz_jpeg_scan: block bits, offsets scan, emit, 0xFF count)
on a q-quantised tiled image, 4:4:4 and 4:2:0; prints the scan size and a hash of its bytes.
Usage: run_scan.py W H [iters]"""
import hashlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
from guetzli_amd.encoder import load_host

w, h = int(sys.argv[1]), int(sys.argv[2])
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 50
L = guetzli_amd.load()
H = load_host()
rgb = images.tiled(w, h)
q = np.full((3, 64), 4, np.int32)
with L.context(rgb, 1.0) as ctx:
    ctx.encode_rgb()
    for frame in ("444", "420"):
        if frame == "420":
            ctx.downsample()
        ctx.quantize(q)
        counts = ctx.jpeg_histograms(q)
        head, depth, code = H.jpeg_head(counts, w, h, q, 3, factor=1 if frame == "444" else 2)
        n = ctx.jpeg_scan(3, depth, code)
        t0 = time.perf_counter()
        for _ in range(iters):
            n = ctx.jpeg_scan(3, depth, code)
        dt = (time.perf_counter() - t0) / iters
        scan = ctx.jpeg_scan_bytes()
        print(f"{w}x{h} {frame}: scan {n} bytes sha256 {hashlib.sha256(scan).hexdigest()[:16]} "
              f"{dt*1e6:.1f} us per gz_jpeg_scan (wall clock, incl. one synchronisation)")
