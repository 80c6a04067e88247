#!/usr/bin/env python3
"""Round trip of small device work through the C ABI with and without PyTorch loaded in the process
(bench.py imports it for torch.cuda.synchronize / torch.distributed): sync_latency.py [torch]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
    torch.cuda.set_device(0); torch.cuda.synchronize()
import numpy as np
import guetzli_amd, images
L = guetzli_amd.load()
with L.context(images.crop(64, 64, 100, 60), 0.97) as ctx:
    ctx.encode_rgb(download=False)
    q = np.full((3, 64), 3, np.int32)
    ctx.quantize(q, download=False)
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(2000): ctx.quantize(q, download=False); ctx.synchronize()
        dt = (time.perf_counter() - t0) / 2000
        t0 = time.perf_counter()
        for _ in range(300): ctx.compare()
        dc = (time.perf_counter() - t0) / 300
        print(sys.argv[1:] or "plain", f"quantize+sync {dt*1e6:.1f} us, 64x64 compare {dc*1e6:.1f} us")
