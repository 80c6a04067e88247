#!/usr/bin/env python3
"""bench.py -- throughput of the Guetzli hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): one 1920x1080 sRGB image (tests/golden/bees.png
tiled from the origin), --quality 95 (butteraugli target 0.971769), one
image per GPU ("weak" scaling: rank r processes the image circularly shifted by
(37r, 53r), as in config 5).

A STEP is one pass of the hot path over that image: one candidate evaluation exactly as
Processor::TryQuantMatrix performs it (processor.cc:298-326) -- ApplyGlobalQuantization
(quantize.h) -> integer IDCT -> YCbCr->RGB -> linear -> full butteraugli distance map and
its maximum -- with the image, its original coefficients and the original's PsychoImage
already resident in HBM; only the 192-entry quant matrix goes in and the 4-byte distance
comes out.  `value` = megapixels of candidate evaluated per second, whole job.

Also reported on the same JSON line:
  roofline     -- HBM roofline of the Compare chain: SURVEY.md 8(d) algorithmic bytes
                  (494 B/px per Compare) / average Compare duration measured with HIP
                  events on the stream the kernels run on.
  cpu_baseline -- the unmodified reference (oracle/_ref, 1 thread) doing the same step on
                  this box's host CPU, rank 0, N=1 only, bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_PX = 494.0          # 123.5 float-plane passes per Compare
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s spec
TARGET_Q95 = 0.971769              # ButteraugliScoreForQuality(95), quality.cc:31-85
W, H = 1920, 1080


def quant_matrix(step):
    """A deterministic family of candidate matrices like QuantMatrixGenerator's
    (processor.cc:269-279): values in {1, 3, 5} growing with frequency."""
    import numpy as np
    k = np.arange(64)
    zz = (k // 8) + (k % 8)
    lvl = 1 + 2 * ((zz + step) % 3 == 0) + 2 * (zz > 6 + step % 4)
    return np.broadcast_to(lvl.astype(np.int32), (3, 64)).copy()


def cpu_baseline(rgb, sample_steps):
    """Reference ButteraugliComparator::Compare + ApplyGlobalQuantization on the host."""
    import numpy as np
    from checkers import ref, oracle
    chk, kind = (ref, "reference") if ref is not None else (oracle, "port")
    h, w, _ = rgb.shape
    co = chk.encode_rgb(rgb)
    cmp_ = chk.comparator(rgb, TARGET_Q95)
    t0 = time.perf_counter()
    for s in range(sample_steps):
        cq, _, _ = chk.reconstruct(co, w, h, quant_matrix(s))
        cmp_.compare(cq)
    dt = time.perf_counter() - t0
    cmp_.close()
    return {"value": round(sample_steps * w * h / 1e6 / dt, 5), "unit": "MPix/s", "cores": 1,
            "kind": kind,
            "sample": f"{sample_steps} candidate evaluations of the same {w}x{h} image, "
                      f"{dt:.1f} s of CPU, single thread ({os.cpu_count()} host cores present)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=12)
    args = ap.parse_args()

    import numpy as np
    import torch
    import guetzli_amd
    import images

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE {world} != --gpus {args.gpus}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))

    L = guetzli_amd.load()
    rgb = images.shifted(images.tiled(W, H), rank)
    ctx = L.context(rgb, TARGET_Q95, device=local_rank)
    ctx.encode_rgb(download=False)

    def step(i):
        ctx.quantize(quant_matrix(i), download=False)
        ctx.compare_enqueue(1)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    fence()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    fence()
    dt = time.perf_counter() - t0
    last = ctx.last_distance()
    assert last > 0.0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline leg: HIP events on the context's own stream around Compare only
    iters = max(10, args.steps)
    ms = ctx.time_compare(iters) / iters
    achieved = ALGO_BYTES_PER_PX * W * H / (ms * 1e-3) / 1e9

    if rank == 0:
        out = {
            "metric": "MPix/s of candidate evaluation at --quality 95 "
                      "(quantize + IDCT + butteraugli Compare; encode search driver not yet on device path)",
            "value": round(world * args.steps * W * H / 1e6 / dt, 3),
            "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64 (butteraugli), int32/int16 (DCT/quantize)",
            "data": "synthetic (tests/golden/bees.png tiled to 1920x1080, SURVEY 8d)",
            "config": {"workload": "single 1920x1080 sRGB image, --quality 95, one candidate "
                                   "evaluation per step, 1 image per GPU",
                       "butteraugli_target": TARGET_Q95, "images_per_gpu": 1},
            "roofline": {"bound": "hbm", "kernel": "Compare chain (all kernels of one gz_compare)",
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None,
                         "ms_per_compare": round(ms, 4),
                         "algorithmic_bytes_per_compare": ALGO_BYTES_PER_PX * W * H},
            "last_distance": last,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(rgb, args.cpu_steps)
        print(json.dumps(out), flush=True)
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
