import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
from checkers import oracle
L = guetzli_amd.load()
w, h = 16, 8
rgb = images.crop(w, h, 300, 150)
oc = oracle.comparator(rgb, 0.971769)
ctx = L.context(rgb, 0.971769)
co = ctx.encode_rgb()
cq = ctx.quantize(np.full((3, 64), 3, np.int32))
for rep in range(2):
    off, idx, err = ctx.block_zeroing_orders()
    print("gpu off", off, "\n idx", idx[:40], "\n err", err[:40])
eoff, eidx, eerr = oc.block_zeroing_orders(cq, co)
print("orc off", eoff, "\n idx", eidx[:40], "\n err", eerr[:40])
# lookahead=1: pure ranking order
for la in (1,):
    off, idx, err = ctx.block_zeroing_orders(lookahead=la)
    eoff, eidx, eerr = oc.block_zeroing_orders(cq, co, lookahead=la)
    print("la1 gpu", idx[:20], err[:20]); print("la1 orc", eidx[:20], eerr[:20])
