#!/usr/bin/env python3
"""bench.py -- MPix/s encoded at --quality 95 on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (BASELINE.json configs[1]): one 1920x1080 sRGB image (tests/golden/bees.png tiled
from the origin), --quality 95 (butteraugli target 0.971769).  One image per
GPU per step ("weak" scaling: rank r encodes the image circularly shifted by (37r, 53r)
pixels, as in config 5); images are independent, so there is no data-path collective --
only the barrier and the max-over-ranks of the elapsed time.

A STEP is one whole encode: guetzli::Process(params, stats, rgb, w, h, &out) through the
host search driver (guetzli_amd/host) with every per-pixel / per-block operation on the GPU
behind the C ABI (RGB->YCbCr + FDCT, quantise, IDCT + colour + butteraugli Compare for
every candidate, the per-block zeroing search, JPEG entropy coding of every candidate).
`value` = megapixels encoded per second, whole job.  The step starts from packed 8-bit RGB
in host memory and ends with the JPEG bytes in host memory, so the (small) PCIe traffic of
the boundary is inside the number.  Rank 0's output is checked against the reference's
JPEG (SHA-256 recorded from the unmodified reference) after the timed region.

Also on the JSON line:
  roofline     -- HBM roofline of the butteraugli evaluation (the second half of the
                  metric): SURVEY.md 8(d) algorithmic bytes of one Compare (494 B/px) /
                  average duration of one Compare chain measured with HIP events on the
                  stream the kernels run on (gz_time_compare), same process, same image.
                  `traffic` = HBM bytes of one chain from the rocprofv3 FETCH_SIZE /
                  WRITE_SIZE passes committed under profiles/ (the counters cannot be read
                  from inside this process).  `roofline_4k` = the same measurement on
                  BASELINE.json configs[2] (3840x2160), the size the chain fills the chip at.
  cpu_baseline -- the unmodified reference guetzli::Process (oracle/_ref, 1 thread) on this
                  box's host CPU, rank 0, N=1 only, on bounded samples: the bench image's top-left
                  640x360, and BASELINE config 0 verbatim (tests/bees.png, --quality 95).
  other_configs.config5_slice -- BASELINE config 5's work split on the GPUs this run has:
                  3840x2160 images, 8 per GPU (image k -> rank k mod N), 8 in flight per GPU,
                  records all-gathered over the process group; every output whose reference
                  hash is committed (tests/golden/config5/) is checked.  `--config5` runs only
                  this leg and reports it as `value`.
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

ALGO_BYTES_PER_PX = 494.0          # 123.5 float-plane passes per Compare
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s HBM3E
QUALITY = 95.0
TARGET_Q95 = 0.971769              # ButteraugliScoreForQuality(95), quality.cc:31-85
W, H = 1920, 1080
GOLDEN_SHA_1080P_Q95 = "9c0eb414b8e73f4372c0b089eafe2350e6ff2ae83926d1c0c5f35cb5f7919729"
GOLDEN_SHA_4K = {95: "481507d21e4d37f296ae6a2a93a84d950c4a135df310b3408a64390b25d59c05",
                 84: "f3be1e4385a977853f7cd224e1722a728c1fa0bc68f1115c27e657c23a028ac0"}
CHAIN = ("butteraugli Compare chain (15 launches per Compare on 3 streams: k_reconstruct, 5 fused "
         "k_blur2d (radius < 16), 3 k_blur_h + 3 k_blur_v (radius >= 16; the mask's radius-20 pair is one "
         "launch per pass), k_malta (both channels), k_mask_pre, k_combine)")


def cpu_baseline():
    """Reference guetzli::Process on the host CPU, bounded sample of the same workload: the
    bench image's top-left 640x360 (bees.png tiled), ~13 s of one core."""
    import images
    from checkers import ref
    if ref is None:
        return None
    rgb = images.tiled(640, 360)
    h, w, _ = rgb.shape
    t0 = time.perf_counter()
    jpg, _ = ref.process(rgb, TARGET_Q95)
    dt = time.perf_counter() - t0
    # BASELINE config 0 verbatim: tests/bees.png --quality 95 on the reference CPU path
    bees = images.bees()
    t0 = time.perf_counter()
    jb, _ = ref.process(bees, TARGET_Q95)
    db = time.perf_counter() - t0
    assert hashlib.sha256(jb).hexdigest() == "f2673f12a4856e020627fa151493a80b1cb2ee4dc81e28afc62dc089baf50242"
    return {"value": round(w * h / 1e6 / dt, 6), "unit": "MPix/s", "cores": 1,
            "kind": "reference",
            "sample": f"unmodified reference guetzli::Process on the top-left {w}x{h} of the "
                      f"bench image, --quality 95: {dt:.1f} s of CPU, single thread "
                      f"({os.cpu_count()} host cores present); output {len(jpg)} bytes",
            "config0": {"workload": "tests/bees.png 444x258 --quality 95 (BASELINE configs[0])",
                        "seconds": round(db, 2), "value": round(444 * 258 / 1e6 / db, 6),
                        "unit": "MPix/s", "output_sha256_matches_golden": True}}


def config5_goldens():
    out = {}
    d = os.path.join(ROOT, "tests", "golden", "config5")
    if os.path.isdir(d):
        for f in os.listdir(d):
            if f.endswith(".json"):
                r = json.load(open(os.path.join(d, f)))
                out[(r["k"], r["w"], r["h"])] = r
    out.setdefault((0, 3840, 2160), {"jpeg_sha256": GOLDEN_SHA_4K[95]})
    return out


def config5_leg(host, images, torch, dist, rank, world, local_rank, images_per_gpu, in_flight, size):
    """8 images per GPU of config 5's batch; returns the JSON object of the leg (rank 0)."""
    from guetzli_amd.batch import run_config5
    w5, h5 = size
    base = images.tiled(w5, h5)
    get = lambda k: images.shifted(base, k)
    proc = lambda im: host.process(im, quality=QUALITY, device=local_rank)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
    run_config5(get, min(2, images_per_gpu), proc, rank, world, dist, in_flight, fence, "cuda")   # warm-up
    recs, secs = run_config5(get, images_per_gpu, proc, rank, world, dist, in_flight, fence, "cuda")
    gold = config5_goldens()
    checked = 0
    for r in recs:
        g = gold.get((r["index"], w5, h5))
        if g is not None:
            assert r["sha256"] == g["jpeg_sha256"], f"config 5 image {r['index']} differs from the reference"
            checked += 1
    n = len(recs)
    return {"workload": f"{n} independent {w5}x{h5} images (the bench image circularly shifted by "
                        f"(37k, 53k)), --quality 95, {images_per_gpu} per GPU (image k -> rank k mod {world}), "
                        f"{in_flight} in flight per GPU; records all-gathered",
            "images": n, "images_per_gpu": images_per_gpu, "in_flight": in_flight,
            "seconds": round(secs, 3), "value": round(n * w5 * h5 / 1e6 / secs, 3), "unit": "MPix/s",
            "outputs_checked_against_reference_hashes": checked,
            "distinct_outputs": len({r["sha256"] for r in recs})}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the 3840x2160 legs (configs[2], [3])")
    ap.add_argument("--batch-images", type=int, default=16,
                    help="images of the extra concurrent-batch leg (0 = skip)")
    ap.add_argument("--batch-workers", type=int, default=8)
    ap.add_argument("--config5", action="store_true",
                    help="run only BASELINE config 5's slice (8 x 4K per GPU) and report it as value")
    ap.add_argument("--images-per-gpu", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=8)
    ap.add_argument("--size", default="4k", choices=["4k", "1080p"])
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 leg of the default run")
    args = ap.parse_args()

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

    host = guetzli_amd.load_host()        # links the gfx950 C-ABI library; no fallback
    size5 = (3840, 2160) if args.size == "4k" else (1920, 1080)
    if args.config5:
        leg = config5_leg(host, images, torch, dist, rank, world, local_rank, args.images_per_gpu,
                          args.in_flight, size5)
        if rank == 0:
            print(json.dumps({
                "metric": "MPix/s encoded at --quality 95", "value": leg["value"], "unit": "MPix/s",
                "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": round(leg["seconds"] * 1e3, 1),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32+f64 (butteraugli), int32/int16 (DCT/quantize/entropy coding)",
                "data": "synthetic (tests/golden/bees.png tiled, circularly shifted per image, SURVEY 8d)",
                "config": leg}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return
    rgb = images.shifted(images.tiled(W, H), rank)

    def step():
        return host.process(rgb, quality=QUALITY, device=local_rank)

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        jpg, info = step()
    fence()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # roofline leg: HIP events on the context's stream around whole Compare chains
    L = guetzli_amd.load()
    with L.context(rgb, TARGET_Q95, device=local_rank) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), 3, np.int32), download=False)
        ctx.time_compare(5)
        iters = 50
        ms = ctx.time_compare(iters) / iters
    achieved = ALGO_BYTES_PER_PX * W * H / (ms * 1e-3) / 1e9
    ms_4k = None
    if rank == 0:
        with L.context(images.tiled(3840, 2160), TARGET_Q95, device=local_rank) as ctx:
            ctx.encode_rgb(download=False)
            ctx.quantize(np.full((3, 64), 3, np.int32), download=False)
            ctx.time_compare(3)
            ms_4k = ctx.time_compare(20) / 20
    # batch leg (BASELINE config 5 in miniature, rank 0): independent images in flight on one
    # GPU at the same time, one host thread each; reported beside `value`, never instead of it
    batch = None
    if rank == 0 and world == 1 and args.batch_images > 0:
        from guetzli_amd.batch import encode_concurrent
        imgs = [images.shifted(images.tiled(W, H), k) for k in range(args.batch_images)]
        proc = lambda im: host.process(im, quality=QUALITY, device=local_rank)
        encode_concurrent(imgs[:args.batch_workers], proc, args.batch_workers)   # warm-up
        torch.cuda.synchronize()
        tb = time.perf_counter()
        outs = encode_concurrent(imgs, proc, args.batch_workers)
        torch.cuda.synchronize()
        tb = time.perf_counter() - tb
        assert hashlib.sha256(outs[0][0]).hexdigest() == GOLDEN_SHA_1080P_Q95
        batch = {"images": args.batch_images, "in_flight": args.batch_workers,
                 "value": round(args.batch_images * W * H / 1e6 / tb, 4), "unit": "MPix/s",
                 "seconds": round(tb, 3),
                 "note": "independent 1920x1080 images, several in flight on ONE GPU (one host "
                         "thread + one device context each); output 0 checked against the "
                         "reference JPEG"}
    # BASELINE configs[2] and [3] (one 3840x2160 image at quality 95 / 84), steady state:
    # one untimed and one timed encode each, beside -- never instead of -- `value`
    other = None
    if rank == 0 and world == 1 and not args.no_4k:
        other = {}
        img4k = images.tiled(3840, 2160)
        for q, golden in ((95, GOLDEN_SHA_4K[95]), (84, GOLDEN_SHA_4K[84])):
            host.process(img4k, quality=q, device=local_rank)
            torch.cuda.synchronize()
            t4 = time.perf_counter()
            j4, i4 = host.process(img4k, quality=q, device=local_rank)
            t4 = time.perf_counter() - t4
            assert hashlib.sha256(j4).hexdigest() == golden
            other[f"3840x2160_q{q}"] = {"seconds": round(t4, 3),
                                        "value": round(3840 * 2160 / 1e6 / t4, 3), "unit": "MPix/s",
                                        "iterations": i4["counters"].get("number of iterations"),
                                        "output_sha256_matches_reference": True}
    c5 = None
    if not args.no_4k and not args.no_config5:   # every rank takes part
        c5 = config5_leg(host, images, torch, dist, rank, world, local_rank, args.images_per_gpu,
                         args.in_flight, size5)
    traffic = {}
    try:
        traffic = json.load(open(os.path.join(ROOT, "profiles", "r02_compare_pmc_traffic.json")))
    except Exception:
        pass

    if rank == 0:
        sha = hashlib.sha256(jpg).hexdigest()
        assert sha == GOLDEN_SHA_1080P_Q95, f"output JPEG differs from the reference: {sha}"
        out = {
            "metric": "MPix/s encoded at --quality 95",
            "value": round(world * args.steps * W * H / 1e6 / dt, 4),
            "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32+f64 (butteraugli), int32/int16 (DCT/quantize/entropy coding)",
            "data": "synthetic (tests/golden/bees.png tiled to 1920x1080, SURVEY 8d)",
            "config": {"workload": "single 1920x1080 sRGB image, --quality 95, whole "
                                   "guetzli::Process per step (host RGB in, JPEG bytes out), "
                                   "1 image per GPU per step",
                       "butteraugli_target": TARGET_Q95, "images_per_gpu": 1,
                       "output_bytes": len(jpg), "output_sha256_matches_reference": True,
                       "iterations": info["counters"].get("number of iterations")},
            "roofline": {"bound": "hbm",
                         "kernel": CHAIN,
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "traffic": traffic.get("1080p", {}).get("traffic_bytes"),
                         "ms_per_compare": round(ms, 4),
                         "algorithmic_bytes_per_compare": ALGO_BYTES_PER_PX * W * H},
            "roofline_4k": {"bound": "hbm", "kernel": CHAIN, "workload": "3840x2160 (configs[2])",
                            "achieved": round(ALGO_BYTES_PER_PX * 3840 * 2160 / (ms_4k * 1e-3) / 1e9, 1),
                            "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                            "frac": round(ALGO_BYTES_PER_PX * 3840 * 2160 / (ms_4k * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                            "traffic": traffic.get("4k", {}).get("traffic_bytes"),
                            "ms_per_compare": round(ms_4k, 4),
                            "algorithmic_bytes_per_compare": ALGO_BYTES_PER_PX * 3840 * 2160},
            # phase A (not HBM-bound -- reported in evaluations, not bytes)
            "block_search": {"evaluations": info["counters"].get("block search evaluations"),
                             "seconds": round(info["timers"].get("block_search", 0.0), 4),
                             "evaluations_per_s": round(info["counters"].get("block search evaluations", 0) /
                                                        max(info["timers"].get("block_search", 0.0), 1e-9)),
                             "note": "CompareBlock evaluations (one 8x8 IDCT + colour + opsin + FFT "
                                     "distance each) of gz_block_zeroing_orders; VALU utilisation of "
                                     "k_block_search: profiles/r02_block_search_pmc.csv"},
            "host_timers_s": {k: round(v, 3) for k, v in info["timers"].items()
                              if k in ("total", "phase_b_host", "compare", "block_search",
                                       "jpeg_write", "create+encode", "select_quant_matrix")},
        }
        if c5 is not None:
            other = dict(other or {})
            other["config5_slice"] = c5
        if other is not None:
            out["other_configs"] = other
        if batch is not None:
            out["batch_one_gpu"] = batch
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()   # rank 0 has the extra legs: leave together
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
