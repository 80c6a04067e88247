#!/usr/bin/env python3
"""Why bench.py's roofline leg sees a slower chain than tools/run_compare.py alone (r4: 1.049 vs
1.004 ms at 4K): the same gz_time_compare before and after whole encodes in one process, with 20 /
100 / 200 iterations, and after a pause.  chain_in_process.py [W H]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import guetzli_amd, images
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
L = guetzli_amd.load(); host = guetzli_amd.load_host()
rgb = images.tiled(w, h)

def chain(tag, iters_list=(20, 100, 200), warm=3):
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), 3, np.int32), download=False)
        ctx.time_compare(warm)
        out = []
        for it in iters_list:
            out.append(f"{it}: {ctx.time_compare(it) / it:.4f}")
        print(f"{tag:34s} ms per Compare  " + "   ".join(out), flush=True)

chain("fresh process")
chain("again (pool warm)")
for k in range(2):
    host.process(rgb, quality=95.0)
chain("after two whole encodes")
chain("same, warm-up 20 chains", warm=20)
time.sleep(1.0)
chain("after a 1 s pause")
from guetzli_amd.batch import encode_concurrent
encode_concurrent([rgb] * 4, lambda im: host.process(im, quality=95.0), workers=4)
chain("after a 4-in-flight batch")
L.lib.gz_trim_pool()
chain("after gz_trim_pool")
