#!/usr/bin/env python3
"""GPU timing of the individual hot-path entry points (wall clock around synchronous C-ABI
calls, i.e. including PCIe transfers of results).  Usage: time_stages.py W H [q]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd  # noqa: E402
import images  # noqa: E402

w, h = int(sys.argv[1]), int(sys.argv[2])
qs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
L = guetzli_amd.load()
rgb = images.tiled(w, h) if max(w, h) > 444 else images.crop(w, h)
t0 = time.perf_counter()
ctx = L.context(rgb, 0.971769)
t1 = time.perf_counter()
co = ctx.encode_rgb()
t2 = time.perf_counter()
cq = ctx.quantize(np.full((3, 64), qs, np.int32))
t3 = time.perf_counter()
d, dm, bm = ctx.compare()
t4 = time.perf_counter()
ms = ctx.time_compare(20) / 20
t5 = time.perf_counter()
off, idx, err = ctx.block_zeroing_orders()
t6 = time.perf_counter()
off, idx, err = ctx.block_zeroing_orders()
t7 = time.perf_counter()
print(f"{w}x{h}: create {t1-t0:.3f}s encode {t2-t1:.4f}s quantize {t3-t2:.4f}s "
      f"compare(sync,+distmap D2H) {t4-t3:.4f}s compare(events) {ms:.3f} ms "
      f"block_search first {t6-t5:.3f}s second {t7-t6:.3f}s "
      f"candidates {off[-1]} ({off[-1]/ctx.nb:.1f}/block) distance {d:.4f}")
