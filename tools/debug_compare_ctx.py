#!/usr/bin/env python3
"""Why is Compare slower inside bench.py than stand-alone?  Times the same chain in several
process states."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
L = guetzli_amd.load()
rgb = images.tiled(1920, 1080)
def t(tag, q=3):
    with L.context(rgb, 0.971769) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), q, np.int32), download=False)
        ctx.time_compare(5)
        ms = ctx.time_compare(50) / 50
        print(f"{tag}: {ms:.4f} ms per Compare", flush=True)
mode = sys.argv[1]
if mode == "plain":
    t("plain"); t("plain again"); t("q=1", 1); t("q=12", 12)
elif mode == "torch":
    import torch
    torch.cuda.init(); torch.cuda.synchronize()
    t("after torch init")
elif mode == "encode":
    host = guetzli_amd.load_host()
    t("before encode")
    host.process(rgb, quality=95)
    t("after one encode")
    t("after one encode again")
