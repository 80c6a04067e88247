#!/usr/bin/env python3
"""Would an incremental Compare pay?  Replays a logged encode on the CPU (tests/replay) with the
shim's dirty-block bookkeeping on and prints, per evaluation, how many 8x8 blocks changed since the
evaluation before and which fraction of the 64x32 tiles lies within the kernels' combined support
(56 px) of a changed block -- what a bit-identical incremental Compare would have to recompute.
Usage: dirty_tiles.py W H QUALITY LOG[.xz]"""
import lzma, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "replay"))
import numpy as np
import build_replay, images
from guetzli_amd.encoder import HostLibrary
w, h, q, log = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4]
tmp = tempfile.mkdtemp()
if log.endswith(".xz"):
    raw = os.path.join(tmp, "log")
    open(raw, "wb").write(lzma.open(log).read())
    log = raw
out = os.path.join(tmp, "dirty.txt")
os.environ.update(GZ_REPLAY_MODE="replay", GZ_REPLAY_FILE=log, GZ_REPLAY_DIRTY_LOG=out)
host = HostLibrary(build_replay.build_host(force=True))
host.process(images.tiled(w, h), quality=q)
rows = np.loadtxt(out, dtype=np.int64)
blocks, tiles = rows[:, 1] / rows[:, 2], rows[:, 3] / rows[:, 4]
print(f"{w}x{h} q{q:g}: {len(rows)} evaluations, {rows[0, 2]} blocks, {rows[0, 4]} tiles of 64x32")
print("evaluation: changed blocks (share), dirty tiles (share)")
for r, b, t in zip(rows, blocks, tiles):
    print(f"  {r[0]:4d}: {r[1]:7d} ({b:6.1%})  {r[3]:6d} ({t:6.1%})")
for name, v in (("changed blocks", blocks), ("dirty tiles", tiles)):
    qs = np.quantile(v, [0.1, 0.25, 0.5, 0.75, 0.9])
    print(f"{name}: min {v.min():.1%}  p10 {qs[0]:.1%}  p25 {qs[1]:.1%}  median {qs[2]:.1%}  p75 {qs[3]:.1%}  p90 {qs[4]:.1%}  mean {v.mean():.1%}")
print("histogram of the dirty-tile share (10 bins of 10 %):", np.histogram(tiles, bins=10, range=(0, 1))[0].tolist())
