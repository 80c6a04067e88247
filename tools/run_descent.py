#!/usr/bin/env python3
"""The device's quick-select descent (k_desc_count / k_desc_swap) on a synthetic order, for
rocprofv3 passes and A/B runs: N entries with keys like phase B's (a few ties), uploaded afresh
before every descent.  run_descent.py [N] [repetitions] [threshold]"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import guetzli_amd, images
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3_200_000
rep = int(sys.argv[2]) if len(sys.argv) > 2 else 10
thr = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
L = guetzli_amd.load()
rng = np.random.default_rng(7)
e = np.zeros(n, L.ORDER_DTYPE if hasattr(L, "ORDER_DTYPE") else np.dtype([("block", np.int32), ("val", np.float32)]))
names = e.dtype.names
kind = sys.argv[4] if len(sys.argv) > 4 else "random"
if kind == "random":
    e[names[0]] = np.arange(n) % 129600
    e[names[1]] = (rng.integers(0, 1 << 20, n) / 64.0).astype(np.float32)
else:
    # like phase B's "down" order: per block a short ascending run (max_err - err[at-1-j]) / weight
    lens = rng.integers(8, 42, n // 20)
    lens = lens[np.cumsum(lens) <= n]
    starts = np.concatenate([[0], np.cumsum(lens)])
    blk = np.repeat(np.arange(len(lens)), lens)
    m = len(blk)
    inrun = np.arange(m) - starts[blk]
    base = rng.random(len(lens)).astype(np.float32) * 3 + 0.5
    wgt = np.array([1.0, 0.5, 1 / 3], np.float32)[rng.integers(0, 3, len(lens))]
    frac = (inrun + rng.random(m)) / lens[blk]
    val = (base[blk] * frac.astype(np.float32)) / wgt[blk]
    if kind == "ties":
        val = np.round(val * 8) / 8
    e = e[:m]
    n = m
    e[names[0]] = blk
    e[names[1]] = val.astype(np.float32)
lib = L.lib
lib.gz_order_descend.restype = C.c_int
lib.gz_order_descend.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p]
with L.context(images.tiled(64, 64), 0.971769) as ctx:
    log = (C.c_uint64 * 36)()
    lv = C.c_int(0)
    times = []
    for _ in range(rep):
        ctx.order_upload(e)
        t0 = time.perf_counter()
        rc = lib.gz_order_descend(ctx.handle, n // 50, thr, 12, log, C.byref(lv))
        times.append(time.perf_counter() - t0)
        assert rc == 0, rc
    cuts = [int(log[3 * i + 2]) for i in range(lv.value)]
    print(f"{n} entries: {lv.value} levels, cuts {cuts}; descent {sorted(times)[len(times)//2]*1e6:.0f} us (host-seen median)")
