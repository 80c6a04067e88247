#!/usr/bin/env python3
"""bench.py -- MPix/s encoded at --quality 95 on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload of `value` (BASELINE.json configs[2], north_star's target size and the rocprof config): one
3840x2160 sRGB image (tests/golden/bees.png tiled from the origin), --quality 95
(butteraugli target 0.971769).  One image per GPU per step ("weak" scaling: rank r encodes the
image circularly shifted by (37r, 53r) pixels, as in config 5); images are independent, so there
is no data-path collective -- only the barrier and the max-over-ranks of the elapsed time.

A STEP is one whole encode: guetzli::Process(params, stats, rgb, w, h, &out) through the
host search driver (guetzli_amd/host) with every per-pixel / per-block operation on the GPU
behind the C ABI (RGB->YCbCr + FDCT, quantise, IDCT + colour + butteraugli Compare for
every candidate, the per-block zeroing search, JPEG entropy coding of every candidate).
`value` = megapixels encoded per second, whole job.  The step starts from packed 8-bit RGB
in host memory and ends with the JPEG bytes in host memory, so the (small) PCIe traffic of
the boundary is inside the number.  Rank 0's output is checked against the reference's
JPEG (SHA-256 recorded from the unmodified reference) after the timed region.
`vs_baseline` = value / BASELINE.md section 2's own measurement of this exact metric on this exact config (C3:
the unmodified reference, one thread of the survey container's Xeon @ 2.10 GHz: 1111.8 s = 0.00746 MPix/s);
`vs_reference_same_box` = value / the same reference on one thread of the bench box's kind of host CPU
(profiles/r05_reference_cpu_4k.json).

Also on the JSON line:
  value_workload -- "3840x2160" (or "1920x1080" with --no-4k): which size `value` is.  `value_4k`
                  and `value_1080p` (+ ms_per_step_4k / _1080p) are ALWAYS emitted under those names,
                  whatever the headline, so that rounds compare like with like (until round 3
                  `value` was the 1080p leg, since round 4 it is the 4K leg: ADVICE r4).
  value_1080p / ms_per_step_1080p / config_1080p / roofline_1080p -- BASELINE configs[1], one
                  1920x1080 image at --quality 95, timed EXACTLY like `value` (same --steps and
                  --warmup, same barrier / synchronise bracket, max over ranks), output hash
                  checked against the reference's.  (Until round 3 this size was `value` and the
                  3840x2160 leg was `value_4k`; --no-4k makes it the headline again.)
  roofline     -- HBM roofline of the butteraugli evaluation (the second half of the
                  metric) on the headline image: SURVEY.md 8(d) algorithmic bytes of one Compare
                  (494 B/px) / average duration of one Compare chain measured with HIP events on
                  the stream the kernels run on (gz_time_compare), same process, same image.  ALWAYS the whole
                  chain, its reconstruction included -- although nine of ten Compares of an encode run without it
                  (gz_config.patch_reconstruct; `compares_4k` counts them over the timed 4K encodes).
                  `traffic` = HBM bytes of one chain from the rocprofv3 FETCH_SIZE /
                  WRITE_SIZE passes committed under profiles/ (the counters cannot be read
                  from inside this process; `traffic_head` = the commit they were taken at);
                  null with `traffic_stale` when the committed figure was measured on other kernel
                  sources than this tree's (tools/gpu_pmc.sh regenerates it).  `valu` inside: the
                  chain's VALU issue time at the two rates an instruction can have (SQ_INSTS_VALU
                  of its kernels from the committed --pmc pass).
  roofline.kernels -- per kernel of the chain: launches per Compare, average duration with the
                  chain serialised on one stream (rocprofv3 --kernel-trace --stats), counter bytes
                  (2 x FETCH_SIZE + WRITE_SIZE), TB/s, VALU wave-instructions; `block_passes`: SURVEY
                  8(d)'s block-pass rooflines (k_reconstruct 18 B/px, k_quantize 12, k_encode_rgb 9).
                  Assembled by tools/kernel_roofline.py from the committed profile set, under the same
                  source-digest guard as `traffic`.
  other_configs.mosaic_3840x2160_q95 -- the same size on content WITHOUT a period (a mosaic of the
                  nine committed photographs, tests/images.mosaic): seconds, MPix/s, iterations,
                  output checked against the reference's hash.  Beside `value`, never instead.
  scale_value  -- BASELINE config 5's work split on the GPUs this run has (the `config5_slice`
                  leg): 3840x2160 images, 8 per GPU (image k -> rank k mod N), 4 in flight per
                  GPU, records all-gathered over the process group; every output whose
                  reference hash is committed (tests/golden/config5/: all 64 images) is checked.
                  The throughput curve of the real multi-GPU workload can be read from it at
                  every N.  `--config5` runs only this leg and reports it as `value`.
  value_1mpix / ms_per_step_1mpix / config_1mpix -- where the tool is mostly used: one 1024x1024 image without a
                  period (tests/images.mosaic), timed exactly like `value`, its output checked against the reference's
                  hash; config_1mpix.batch: 64 such images on one GPU, six in flight; iteration_floor_us: a 64x64
                  image's time per phase-B iteration = the size-independent latency of one iteration.
  first_encode_s -- the process's first encode (HIP start-up, code-object load, pool fill); `value` is steady
                  state and excludes it (and the first 4K encode, first_encode_4k_s: pool growth from 1080p buffers).
  box          -- three-second calibration of the box the run landed on (tools/ubench/bw:
                  streaming copy rates), so that a slow box is visible as such.
  cpu_baseline -- the unmodified reference guetzli::Process (oracle/_ref, 1 thread) on this
                  box's host CPU, rank 0, N=1 only, on bounded samples: the bench image's top-left
                  640x360, and BASELINE config 0 verbatim (tests/bees.png, --quality 95).  `all_cores`: the same
                  sample as one reference PROCESS per usable host CPU (affinity mask capped by the container's CPU
                  quota), all at once -- the reference's own batch form (tests/golden_test.sh:24-26), the figure the
                  per-GPU batch numbers stand beside (tools/ref_cpu_all_cores.py).
  config5_slice.binding_per_rank -- every rank's host-CPU binding (guetzli_amd/affinity.py: the CPUs of its GPU's
                  NUMA node, its share among the ranks on that node).

`--emulate` (CPU dry run, used by tests/test_bench_main.py at world size 2): the same control
flow -- process group, barriers, max-over-ranks, all-gathers, rank-0-only legs -- over gloo
and the test-suite's CPU emulation of the kernels on tiny images.  It measures nothing.
"""
import argparse
import csv
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

BASELINE_MD_MPIX_S = {"3840x2160": 0.00746, "1920x1080": 0.00770}
ALGO_BYTES_PER_PX = 494.0          # 123.5 float-plane passes per Compare
HBM_PEAK_GBPS = 8000.0             # MI355X_MICROARCH.md: 8 TB/s HBM3E
QUALITY = 95.0
TARGET_Q95 = 0.971769              # ButteraugliScoreForQuality(95), quality.cc:31-85
GOLDEN_SHA_1080P_Q95 = "9c0eb414b8e73f4372c0b089eafe2350e6ff2ae83926d1c0c5f35cb5f7919729"
GOLDEN_SHA_4K = {95: "481507d21e4d37f296ae6a2a93a84d950c4a135df310b3408a64390b25d59c05",
                 84: "f3be1e4385a977853f7cd224e1722a728c1fa0bc68f1115c27e657c23a028ac0"}
CHAIN = ("butteraugli Compare chain (17 launches per Compare on 3 streams: k_reconstruct, 5 fused "
         "k_blur2d (radius < 16), 4 k_blur_h + 4 k_blur_v (radius >= 16: LF X/Y, LF B, SameNoise, the mask's "
         "radius-20 pair as one launch per pass), k_malta_rolled (both channels), k_mask_pre, k_combine)")
TRAFFIC_JSONS = [os.path.join(ROOT, "profiles", n) for n in
                 ("r06_compare_pmc_traffic.json", "r05_compare_pmc_traffic.json", "r04_compare_pmc_traffic.json")]


def load_traffic():
    """The committed PMC traffic figures, or {"stale": True, ...} when they were measured on other
    kernel sources than this tree's (guetzli_amd.build.csrc_digest; tools/gpu_pmc.sh regenerates
    them): a figure that no longer describes the code must not sit beside a live measurement."""
    from guetzli_amd.build import csrc_digest
    for path in TRAFFIC_JSONS:
        try:
            t = json.load(open(path))
        except Exception:
            continue
        t.setdefault("source", os.path.relpath(path, ROOT))
        t["stale"] = t.get("csrc_sha256") != csrc_digest()
        return t
    return {"stale": True}


def load_kernel_roofline():
    """profiles/r05_compare_kernels.json (tools/kernel_roofline.py on the GPU box, from the same
    session's rocprofv3 CSVs): the chain's kernels one by one and the block passes.  None when it
    was assembled from other kernel sources than this tree's."""
    from guetzli_amd.build import csrc_digest
    for name in ("r06_compare_kernels.json", "r05_compare_kernels.json"):
        try:
            t = json.load(open(os.path.join(ROOT, "profiles", name)))
        except Exception:
            continue
        if t.get("csrc_sha256") != csrc_digest():
            return {"stale": True, "source": "profiles/" + name, "head": t.get("head")}
        t["source"] = "profiles/" + name
        return t
    return None


def mosaic_golden(name="mosaic_3840x2160_q95"):
    try:
        return json.load(open(os.path.join(ROOT, "tests", "golden", "photos", name + ".json")))
    except Exception:
        return None


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
    # whole BASELINE images through the same reference build on this kind of box, once per round
    # (tools/ref_cpu_time.py under gpurun: 9 and 2.3 minutes -- too long for this line)
    full = {}
    for key, names in (("3840x2160_q95", ("r06_reference_cpu_4k.json", "r05_reference_cpu_4k.json", "r04_reference_cpu_4k.json")),
                       ("1920x1080_q95", ("r06_reference_cpu_1080p.json", "r05_reference_cpu_1080p.json", "r04_reference_cpu_1080p.json"))):
        try:
            name = next(n for n in names if os.path.exists(os.path.join(ROOT, "profiles", n)))
            r = json.load(open(os.path.join(ROOT, "profiles", name)))
            full[key] = {k: r[k] for k in ("seconds", "value", "unit", "host_cpu", "host_cores_present", "cores_used",
                                           "output_sha256", "head")}
            full[key]["source"] = "profiles/" + name
        except Exception:
            pass
    # ... and every host core at once, the reference's own batch form (tests/golden_test.sh:24-26: one process
    # per image under xargs -P): the figure the per-GPU batch numbers (`scale_value`, `batch_*`) stand beside
    all_cores = None
    try:
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ref_cpu_all_cores.py"), str(w), str(h)],
                           capture_output=True, text=True, timeout=240)
        all_cores = json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:
        all_cores = {"error": f"{type(e).__name__}: {e}"}
    return {"value": round(w * h / 1e6 / dt, 6), "unit": "MPix/s", "cores": 1,
            "kind": "reference", "all_cores": all_cores, "full_images_same_box_kind": full,
            "note": "full_images_same_box_kind: the BASELINE images of `value` / `value_1080p` through the unmodified "
                    "reference on the bench box's kind of host CPU, one thread -- COMMITTED records of a gpurun "
                    "session (each carries its commit in `head` and its seconds; 9 and 2.3 minutes are too long for "
                    "this line), output hashes equal to the GPU's; timed in THIS run: the bounded sample below",
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


def block_search_counters():
    """SQ counters of k_block_search from the committed rocprofv3 --pmc pass (tools/gpu_pmc.sh; the
    counters cannot be read from inside this process): the share of its wave cycles in which a
    wavefront issues a VALU instruction, and VALU instructions per wavefront."""
    import csv
    for name in ("r06_block_search_pmc.csv", "r05_block_search_pmc.csv", "r04_block_search_pmc.csv", "r03_block_search_pmc.csv", "r02_block_search_pmc.csv"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        c = {}
        import re
        m = re.search(r"#\s*evaluations per launch\s+(\d+)", open(path).read())
        evals = int(m.group(1)) if m else None
        for r in csv.DictReader(l for l in open(path) if not l.startswith("#")):   # (first lines: the commit stamp, the evaluations)
            if "k_block_search<0>" in r["kernel"]:
                c[r["counter"]] = float(r["avg_value"])
        if "SQ_WAVE_CYCLES" in c and "SQ_ACTIVE_INST_VALU" in c:
            return {"source": "profiles/" + name,
                    "valu_active_per_wave_cycle": round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4),
                    "wait_any_per_wave_cycle": round(c.get("SQ_WAIT_ANY", 0.0) / c["SQ_WAVE_CYCLES"], 4),
                    "valu_wave_instructions_per_launch": c.get("SQ_INSTS_VALU"),
                    "evaluations_per_launch": evals,
                    "valu_wave_instructions_per_evaluation":
                        round(c["SQ_INSTS_VALU"] / evals, 1) if evals and c.get("SQ_INSTS_VALU") else None}
    return None


def box_calibration():
    """Streaming copy rates of this box (tools/ubench/bw, < 1 s): the Compare chain follows them."""
    exe = os.path.join(ROOT, "tools", "ubench", "bw")
    if not os.path.exists(exe):
        return None
    try:
        out = subprocess.run([exe], capture_output=True, text=True, timeout=60).stdout
    except Exception:
        return None
    import re
    cal = {}
    for line in out.splitlines():
        m = re.match(r"^copy (.+?)\s+([\d.]+) us/plane\s+([\d.]+) TB/s", line)
        if m:
            cal["copy_" + m.group(1).strip().replace(" ", "_") + "_TBps"] = float(m.group(3))
    return cal or None


class Env:
    """Device and process-group plumbing: MI355X + RCCL, or (--emulate) the CPU dry run."""

    def __init__(self, args):
        import torch
        self.torch = torch
        self.emulate = args.emulate
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        assert self.world == args.gpus, f"WORLD_SIZE {self.world} != --gpus {args.gpus}"
        self.dist = None
        self.cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else os.cpu_count()
        if not self.emulate:
            assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback exists)"
            torch.cuda.set_device(self.local_rank)
        self.tensor_device = "cpu" if self.emulate else "cuda"
        self.device = 0 if self.emulate else self.local_rank   # device index of this rank's contexts
        if self.world > 1:
            self.bind_cpus()
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if self.emulate:
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
            else:
                dist.init_process_group("nccl", rank=self.rank, world_size=self.world,
                                        device_id=torch.device("cuda", self.local_rank))
            self.dist = dist

    def bind_cpus(self):
        """One process per GPU: rank r keeps to the host cores of GPU r's NUMA node -- its share of them among the
        ranks whose GPUs hang off the same node, whole physical cores (guetzli_amd/affinity.py: sysfs numa_node /
        local_cpulist of the GPU's PCI function; contiguous shares of the allowed cores where sysfs says nothing).
        Its image threads, code-refresh helpers and the driver's worker pool then stay next to its GPU and off the
        other ranks' cores.  BENCH_NO_AFFINITY=1 leaves the scheduler alone.  The binding is on the JSON line
        (`binding_per_rank`)."""
        from guetzli_amd import affinity
        self.binding = {"rank": self.rank, "how": "not bound"}
        if os.environ.get("BENCH_NO_AFFINITY") or not hasattr(os, "sched_setaffinity"):
            return
        try:
            allowed = sorted(os.sched_getaffinity(0))
            if self.emulate:
                bus_ids = [None] * self.world
            else:
                import guetzli_amd
                bus_ids = affinity.device_bus_ids(self.world, guetzli_amd.load())
            p = affinity.plan(self.local_rank, self.world, bus_ids, allowed,
                              os.environ.get("BENCH_SYSFS_ROOT", "/sys"))
            if p["cpus"]:
                os.sched_setaffinity(0, p["cpus"])
            else:
                print(f"bench.py: rank {self.rank}: {p['how']}", file=sys.stderr, flush=True)
            self.binding = {"rank": self.rank, "gpu_pci_bus_id": p["gpu_pci_bus_id"], "numa_node": p["numa_node"],
                            "host_cpus": affinity.format_cpulist(p["cpus"] or allowed),
                            "n_host_cpus": len(p["cpus"] or allowed), "ranks_on_node": p["ranks_on_node"],
                            "how": p["how"]}
        except OSError as e:
            print(f"bench.py: rank {self.rank}: sched_setaffinity failed ({e}) -- not binding",
                  file=sys.stderr, flush=True)
        self.cores = len(os.sched_getaffinity(0))

    def bindings(self):
        """Every rank's binding record, on every rank (all-gathered; one record at world size 1)."""
        mine = getattr(self, "binding", {"rank": self.rank, "how": "not bound (single process)",
                                         "n_host_cpus": self.cores})
        if self.dist is None:
            return [mine]
        got = [None] * self.world
        self.dist.all_gather_object(got, mine)
        return got

    def sync(self):
        if not self.emulate:
            self.torch.cuda.synchronize()

    def fence(self):
        self.sync()
        if self.dist is not None:
            self.dist.barrier()
        self.sync()

    def max_over_ranks(self, seconds):
        if self.dist is None:
            return seconds
        t = self.torch.tensor([seconds], dtype=self.torch.float64, device=self.tensor_device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def libraries(self):
        """(host driver, C-ABI library): the gfx950 build, or the emulation build for the dry run."""
        import guetzli_amd
        if self.emulate:
            sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
            import build_emu
            from guetzli_amd.capi import Library
            from guetzli_amd.encoder import HostLibrary
            if self.rank == 0:
                build_emu.build_host()
            if self.dist is not None:
                self.dist.barrier()
            return HostLibrary(build_emu.HOST_LIB), Library(build_emu.LIB)
        return guetzli_amd.load_host(), guetzli_amd.load()   # no fallback: fails without the library

    def finish(self):
        if self.dist is not None:
            self.dist.barrier()   # rank 0 has the extra legs: leave together
            self.dist.destroy_process_group()


def timed_steps(env, step, steps, warmup):
    """W untimed steps, then exactly K steps between barrier + synchronise on both sides; the
    maximum over ranks.  Returns (seconds, last result)."""
    for _ in range(warmup):
        step()
    env.fence()
    t0 = time.perf_counter()
    res = None
    del STEP_MS[:]
    for _ in range(steps):
        ts = time.perf_counter()
        res = step()
        STEP_MS.append(round((time.perf_counter() - ts) * 1e3, 2))   # (this rank's; no extra synchronisation)
    env.fence()
    return env.max_over_ranks(time.perf_counter() - t0), res


STEP_MS = []   # the last timed_steps' individual steps on this rank (reported beside ms_per_step)


def ranks_seen(env):
    """Sum over the process group of 1 per rank (RCCL all-reduce on the GPUs; gloo in the dry run)."""
    if env.dist is None:
        return 1
    t = env.torch.ones(1, dtype=env.torch.int32, device=env.tensor_device)
    env.dist.all_reduce(t, op=env.dist.ReduceOp.SUM)
    return int(t.item())


def config5_leg(env, host, images, images_per_gpu, in_flight, size, quality, from_png=False):
    """8 images per GPU of config 5's batch; returns the JSON object of the leg (rank 0).
    from_png: the batch as configs[4] words it -- PNG files: every image arrives as PNG bytes
    (encoded once, outside the timed region) and is decoded inside it by the product's reader
    (guetzli_amd::ReadPng = the reference front end's ReadPNG, guetzli.cc:47-152) on host threads
    beside the other images' device work, ahead of the image's turn on the GPU (batch.py, prepare)."""
    from guetzli_amd.batch import run_config5
    w5, h5 = size
    base = images.tiled(w5, h5)
    if from_png:
        import io
        from PIL import Image
        pngs = {}

        def get(k):
            if k not in pngs:
                b = io.BytesIO()
                Image.fromarray(images.shifted(base, k)).save(b, "PNG", compress_level=1)
                pngs[k] = b.getvalue()
            return pngs[k]
        for k in range(env.rank, images_per_gpu * env.world, env.world):
            get(k)   # (encoded before anything is timed)
        # (decoded ahead of the image's turn on the GPU, on threads of their own: guetzli_amd/batch.py)
        prep = host.read_png
        proc = lambda im: host.process(im, quality=quality, device=env.device)
    else:
        prep = None
        # this rank's images exist before anything is timed (as the PNG bytes do above): the timed region
        # is the encodes, not numpy's circular shifts of a 25 MB array on the image threads
        mine = {k: images.shifted(base, k) for k in range(env.rank, images_per_gpu * env.world, env.world)}
        get = lambda k: mine[k] if k in mine else images.shifted(base, k)
        proc = lambda im: host.process(im, quality=quality, device=env.device)
    if not env.emulate:
        run_config5(get, min(2, images_per_gpu), proc, env.rank, env.world, env.dist, in_flight, env.fence,
                    env.tensor_device, prepare=prep)   # warm-up
    recs, secs = run_config5(get, images_per_gpu, proc, env.rank, env.world, env.dist, in_flight,
                             env.fence, env.tensor_device, prepare=prep)
    gold = {} if env.emulate else config5_goldens()
    checked = 0
    for r in recs:
        g = gold.get((r["index"], w5, h5))
        if g is not None:
            assert r["sha256"] == g["jpeg_sha256"], f"config 5 image {r['index']} differs from the reference"
            checked += 1
    n = len(recs)
    return {"workload": f"{n} independent {w5}x{h5} images (the bench image circularly shifted by "
                        f"(37k, 53k)){' as PNG bytes, decoded inside the timed region' if from_png else ''}, "
                        f"--quality {quality:g}, {images_per_gpu} per GPU (image k -> rank k mod "
                        f"{env.world}), {in_flight} in flight per GPU; records all-gathered",
            "images": n, "images_per_gpu": images_per_gpu, "in_flight": in_flight,
            "seconds": round(secs, 3), "value": round(n * w5 * h5 / 1e6 / secs, 3), "unit": "MPix/s",
            "outputs_checked_against_reference_hashes": checked,
            "distinct_outputs": len({r["sha256"] for r in recs}),
            # what the process group saw: an all-reduced count of the ranks, and from the gathered
            # records the ranks that delivered images with each rank's busiest image thread
            "n_ranks_seen": ranks_seen(env),
            "ranks_in_records": sorted({r["rank"] for r in recs}),
            "images_per_rank": {str(k): sum(1 for r in recs if r["rank"] == k)
                                for k in sorted({r["rank"] for r in recs})},
            "encode_seconds_per_rank": {str(k): round(sum(r["seconds"] for r in recs if r["rank"] == k), 3)
                                        for k in sorted({r["rank"] for r in recs})},
            "host_cores_per_rank": env.cores, "binding_per_rank": env.bindings()}


def _latest_profile(name):
    """profiles/<round>_<name> of the newest round that has it."""
    for rn in ("r06", "r05", "r04", "r03", "r02"):
        path = os.path.join(ROOT, "profiles", f"{rn}_{name}")
        if os.path.exists(path):
            return path
    return os.path.join(ROOT, "profiles", "r04_" + name)


SQ_COUNTERS = {"1080p": _latest_profile("compare_1080p_sq_counters.csv"),
               "4k": _latest_profile("compare_4k_sq_counters.csv")}
# wave64 VALU instructions the chip issues per second: f32 multiplies / adds go at one per 2 clocks
# and SIMD once a SIMD holds two or more waves (tools/ubench/pk.hip: 73 T lane-ops/s measured),
# FP64 and -- as far as this repo has measured anything -- the rest at one per 4 clocks
VALU_WAVE_INSTR_PER_S_F32 = 256 * 4 * 0.5 * 2.4e9
VALU_WAVE_INSTR_PER_S_4CLK = 256 * 4 * 0.25 * 2.4e9
NOT_IN_CHAIN = ("k_encode_rgb", "k_linear_from_rgb8", "k_quantize", "__amd_rocclr")


def valu_floor(size, ms_measured):
    """The chain's VALU issue time: SQ_INSTS_VALU of its kernels (rocprofv3 --pmc, committed under
    profiles/ -- the counters cannot be read from inside this process) per Compare / the chip's
    VALU issue rate, at both rates an instruction can have (the counters do not split the
    instructions by class, so the chain's real floor lies between the two).  With FMA contraction
    off (the reference is SSE2) every multiply and add is an instruction of its own.  A kernel's
    count is its average per launch x its launches per Compare: the profiled run also launches the
    opsin / LF / MF / HF kernels once for the ORIGINAL image when the context is created, and that
    launch is not part of a Compare (until round 3's last session it was counted as a third of one)."""
    try:
        rows = [r for r in csv.reader(l for l in open(SQ_COUNTERS[size]) if not l.startswith("#"))][1:]
        valu = {r[0]: (int(r[2]), float(r[3])) for r in rows if r[1] == "SQ_INSTS_VALU"}
        compares = valu["gz::k_combine"][0]          # one k_combine per Compare
        per_kernel = {k: max(1, n // compares) * v for k, (n, v) in valu.items()
                      if not any(x in k for x in NOT_IN_CHAIN)}
        total = sum(per_kernel.values())
        lo_ms = total / VALU_WAVE_INSTR_PER_S_F32 * 1e3
        hi_ms = total / VALU_WAVE_INSTR_PER_S_4CLK * 1e3
        top = max(per_kernel, key=per_kernel.get)
        return {"wave_instructions_per_compare": round(total),
                "floor_ms_f32_rate": round(lo_ms, 4), "floor_ms_4clk_rate": round(hi_ms, 4),
                "frac_of_measured_f32_rate": round(lo_ms / ms_measured, 4),
                "frac_of_measured_4clk_rate": round(hi_ms / ms_measured, 4),
                "largest": {"kernel": top.replace("gz::", ""), "share": round(per_kernel[top] / total, 3)},
                "source": os.path.relpath(SQ_COUNTERS[size], ROOT),
                "note": "sum over the chain's kernels of SQ_INSTS_VALU per launch x launches per Compare, / "
                        "(1024 SIMDs x 2.4 GHz / 2 clocks [f32 mul/add rate, tools/ubench/pk.hip] or / 4 clocks "
                        "[FP64 and the rest]); the chain's VALU floor lies between the two"}
    except Exception as e:   # (profiles not present: the line stays valid)
        return {"error": str(e)}


def roofline_of(L, rgb, device, iters, warm):
    w, h = rgb.shape[1], rgb.shape[0]
    with L.context(rgb, TARGET_Q95, device=device) as ctx:
        ctx.encode_rgb(download=False)
        ctx.quantize(np.full((3, 64), 3, np.int32), download=False)
        ctx.time_compare(warm)
        ms = ctx.time_compare(iters) / iters
    achieved = ALGO_BYTES_PER_PX * w * h / (ms * 1e-3) / 1e9
    return ms, achieved


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-4k", action="store_true", help="skip the 3840x2160 legs (configs[2], [3])")
    ap.add_argument("--batch-images", type=int, default=16,
                    help="images of the extra concurrent-batch leg (0 = skip)")
    ap.add_argument("--batch-workers", type=int, default=6,
                    help="images in flight in the 1080p batch leg (6: +7 %% over 4 on the final sources, "
                         "profiles/r06_chain_experiments.log section 19; at 4K 4, 5 and 6 are equal)")
    ap.add_argument("--no-1mpix", action="store_true", help="skip the 1024x1024 legs (value_1mpix, its batch, iteration floor)")
    ap.add_argument("--batch-1mpix", type=int, default=64, help="images of the 1 MPix batch leg (0 = skip)")
    ap.add_argument("--batch-workers-1mpix", type=int, default=6,
                    help="images in flight in the 1 MPix batch leg (6: +12 %% over 4 at this size, profiles/r06_chain_experiments.log)")
    ap.add_argument("--config5", action="store_true",
                    help="run only BASELINE config 5's slice (8 x 4K per GPU) and report it as value")
    ap.add_argument("--images-per-gpu", type=int, default=8)
    ap.add_argument("--in-flight", type=int, default=4,
                    help="images in flight per GPU in the config-5 legs (4 keeps the GPU as busy as 8 or 12 "
                         "do and varies less: profiles/r03_chain_kernel_experiments.log)")
    ap.add_argument("--size", default="4k", choices=["4k", "1080p"])
    ap.add_argument("--no-config5", action="store_true", help="skip the config-5 legs of the default run")
    ap.add_argument("--png", action="store_true", help="with --config5: the images arrive as PNG bytes")
    ap.add_argument("--emulate", action="store_true",
                    help="CPU dry run of the control flow (gloo + the test-suite's emulation of the "
                         "kernels, tiny images): measures nothing")
    args = ap.parse_args()

    env = Env(args)
    import images
    host, L = env.libraries()
    rank, world, local_rank = env.rank, env.world, env.device
    emu = env.emulate
    W, H = (40, 32) if emu else (1920, 1080)
    W4, H4 = (48, 40) if emu else (3840, 2160)
    quality = 84.0 if emu else QUALITY   # (the dry run: the shortest search)
    size5 = (W4, H4) if args.size == "4k" else (W, H)
    if emu:
        args.images_per_gpu = min(args.images_per_gpu, 2)
        args.in_flight = 1            # the emulation of the kernels is single-threaded
        args.batch_images = min(args.batch_images, 2)
        args.batch_workers = 1
    if args.config5:
        leg = config5_leg(env, host, images, args.images_per_gpu, args.in_flight, size5, quality,
                          from_png=args.png)
        if rank == 0:
            print(json.dumps({
                "metric": "MPix/s encoded at --quality 95", "value": leg["value"], "unit": "MPix/s",
                "n_gpus": world, "steps": 1, "warmup": 1, "ms_per_step": round(leg["seconds"] * 1e3, 1),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                "dtype": "f32+f64 (butteraugli), int32/int16 (DCT/quantize/entropy coding)",
                "data": "synthetic (tests/golden/bees.png tiled, circularly shifted per image, SURVEY 8d)",
                "config": leg}), flush=True)
        env.finish()
        return
    rgb = images.shifted(images.tiled(W, H), rank)
    rgb4k = images.shifted(images.tiled(W4, H4), rank)

    def step():
        return host.process(rgb, quality=quality, device=local_rank)

    def step4k():
        return host.process(rgb4k, quality=quality, device=local_rank)

    # the process's first encode, on its own: HIP start-up, code-object load, first allocations
    t0 = time.perf_counter()
    step()
    env.sync()
    first_encode_s = time.perf_counter() - t0

    dt, (jpg, info) = timed_steps(env, step, args.steps, args.warmup)
    step_ms_1080p = list(STEP_MS)
    dt4k = jpg4k = info4k = step_ms_4k = first_encode_4k_s = compares_4k = None
    if not args.no_4k:
        # the process's first encode at this size, on its own like the 1080p one above: the pools grow from
        # 1080p to 4K buffers (and, in a process with PyTorch loaded, the context of the encode after it
        # still takes 20 ms to create where a plain process takes 2: with --warmup 1 that step used to be
        # the first TIMED one, 267 ms among 246 -- `step_ms_4k` shows every timed step)
        t0 = time.perf_counter()
        step4k()
        env.sync()
        first_encode_4k_s = time.perf_counter() - t0
        cc0 = L.compare_counters(all=True)
        dt4k, (jpg4k, info4k) = timed_steps(env, step4k, args.steps, args.warmup)
        step_ms_4k = list(STEP_MS)
        # how many of the encodes' Compares ran without their full reconstruction (gz_config.patch_reconstruct: the
        # calls that change the candidate keep its linear planes current); `roofline` always times the whole chain
        # (... and, for a lone context, their opsin image: gz_config.opsin_ahead)
        cc1 = L.compare_counters(all=True)
        compares_4k = {"compares": cc1[2] - cc0[2], "without_full_reconstruction": cc1[0] - cc0[0],
                       "opsin_image_in_place": cc1[3] - cc0[3]}

    # Where the tool is mostly used: <= 2 MPix.  One 1024x1024 image without a period (tests/images.mosaic), timed
    # exactly like `value` (every rank, same bracket), and -- rank 0 -- 64 of them (circular shifts) four in flight;
    # `iteration_floor_us`: a 64x64 image's time per phase-B iteration = the size-independent latency of one
    # iteration (launches, host round trips), which a 1 MPix image pays 150 times.
    W1, H1 = (32, 32) if emu else (1024, 1024)
    gold1m = None if emu else mosaic_golden("mosaic_1024x1024_q95")
    small = None
    if not args.no_1mpix:
        rgb1m = images.tiled(W1, H1) if emu else images.mosaic(W1, H1)
        step1m = lambda: host.process(rgb1m, quality=quality, device=local_rank)
        step1m()
        dt1m, (jpg1m, info1m) = timed_steps(env, step1m, args.steps, args.warmup)
        small = {"workload": f"single {W1}x{H1} image without a period (tests/images.mosaic), --quality {quality:g}, whole "
                             f"guetzli::Process per step, timed like `value` ({args.steps} steps after {args.warmup} warm-up)",
                 "value": round(world * args.steps * W1 * H1 / 1e6 / dt1m, 4), "unit": "MPix/s",
                 "ms_per_step": round(dt1m / args.steps * 1e3, 2), "step_ms": list(STEP_MS),
                 "iterations": info1m["counters"].get("number of iterations"), "output_bytes": len(jpg1m),
                 "host_timers_s": {k: round(v, 3) for k, v in info1m["timers"].items()
                                   if k in ("total", "phase_b_host", "compare", "block_search", "select_quant_matrix")}}
        if gold1m is not None and rank == 0:
            assert hashlib.sha256(rgb1m.tobytes()).hexdigest() == gold1m["rgb_sha256"], "1 MPix input differs from the golden's"
            assert hashlib.sha256(jpg1m).hexdigest() == gold1m["jpeg_sha256"], "1 MPix output differs from the reference"
            small["output_sha256_matches_reference"] = True
            small["reference_cpu_seconds_build_container"] = gold1m.get("reference_cpu_seconds")
        if rank == 0 and world == 1:
            from guetzli_amd.batch import encode_concurrent
            n1 = 2 if emu else args.batch_1mpix
            if n1 > 0:
                imgs1 = [images.shifted(rgb1m, k) for k in range(n1)]
                proc1 = lambda im: host.process(im, quality=quality, device=local_rank)
                wk1 = 1 if emu else args.batch_workers_1mpix
                encode_concurrent(imgs1[:wk1], proc1, wk1)   # warm-up
                env.sync()
                tb = time.perf_counter()
                outs1 = encode_concurrent(imgs1, proc1, wk1)
                env.sync()
                tb = time.perf_counter() - tb
                assert gold1m is None or hashlib.sha256(outs1[0][0]).hexdigest() == gold1m["jpeg_sha256"]
                small["batch"] = {"images": n1, "in_flight": wk1, "seconds": round(tb, 3),
                                  "value": round(n1 * W1 * H1 / 1e6 / tb, 3), "unit": "MPix/s",
                                  "distinct_outputs": len({hashlib.sha256(o[0]).hexdigest() for o in outs1}),
                                  "note": f"{n1} independent {W1}x{H1} images (the image above circularly shifted by "
                                          "(37k, 53k)), several in flight on ONE GPU; output 0 checked against the reference"}
            # the per-iteration floor
            tiny = images.crop(64, 64, 220, 120) if not emu else images.tiled(16, 16)
            host.process(tiny, quality=quality, device=local_rank)
            runs = []
            for _ in range(5):
                t0 = time.perf_counter()
                _, itiny = host.process(tiny, quality=quality, device=local_rank)
                runs.append(time.perf_counter() - t0)
            it = max(1, int(itiny["counters"].get("number of iterations", 1)))
            small["iteration_floor_us"] = round(sorted(runs)[2] / it * 1e6, 1)
            small["iteration_floor_note"] = (f"a 64x64 image's whole encode / its {it} phase-B iterations (median of five): "
                                             "launch + round-trip latency of one iteration, independent of the image size")

    # roofline legs: HIP events on the context's stream around whole Compare chains
    # (warm-up: the clocks need ~20 ms of sustained chains to settle -- the first 20 chains after an
    # idle moment run 7 % slower than the 200 behind them, tools/chain_in_process.py,
    # profiles/r05_chain_in_process.log; an encode runs 150 of them back to back)
    ms, achieved = roofline_of(L, rgb, local_rank, 2 if emu else 200, 1 if emu else 60)
    ms_4k = achieved_4k = None
    if rank == 0:
        ms_4k, achieved_4k = roofline_of(L, images.tiled(W4, H4), local_rank, 1 if emu else 100, 1 if emu else 20)
    # batch leg (BASELINE config 5 in miniature, rank 0): independent images in flight on one
    # GPU at the same time, one host thread each; reported beside `value`, never instead of it
    batch = None
    if rank == 0 and world == 1 and args.batch_images > 0:
        from guetzli_amd.batch import encode_concurrent
        imgs = [images.shifted(images.tiled(W, H), k) for k in range(args.batch_images)]
        proc = lambda im: host.process(im, quality=quality, device=local_rank)
        encode_concurrent(imgs[:args.batch_workers], proc, args.batch_workers)   # warm-up
        env.sync()
        tb = time.perf_counter()
        outs = encode_concurrent(imgs, proc, args.batch_workers)
        env.sync()
        tb = time.perf_counter() - tb
        assert emu or hashlib.sha256(outs[0][0]).hexdigest() == GOLDEN_SHA_1080P_Q95
        batch = {"images": args.batch_images, "in_flight": args.batch_workers,
                 "value": round(args.batch_images * W * H / 1e6 / tb, 4), "unit": "MPix/s",
                 "seconds": round(tb, 3),
                 "note": f"independent {W}x{H} images, several in flight on ONE GPU (one host "
                         "thread + one device context each); output 0 checked against the "
                         "reference JPEG"}
    # BASELINE configs[3] (one 3840x2160 image at quality 84), steady state: one untimed and one
    # timed encode, beside -- never instead of -- `value`
    other = None
    if rank == 0 and world == 1 and not args.no_4k and not emu:
        other = {}
        img4k = images.tiled(W4, H4)
        host.process(img4k, quality=84, device=local_rank)
        env.sync()
        runs = []
        for _ in range(3):   # (median of three timed encodes: a single one varies by 20 %)
            t4 = time.perf_counter()
            j4, i4 = host.process(img4k, quality=84, device=local_rank)
            runs.append(time.perf_counter() - t4)
        t4 = sorted(runs)[1]
        assert hashlib.sha256(j4).hexdigest() == GOLDEN_SHA_4K[84]
        other["3840x2160_q84"] = {"seconds": round(t4, 3), "value": round(W4 * H4 / 1e6 / t4, 3),
                                  "unit": "MPix/s",
                                  "iterations": i4["counters"].get("number of iterations"),
                                  "output_sha256_matches_reference": True}
        # the headline's size on content without a period: beside `value`
        gold = mosaic_golden()
        if gold is not None:
            mos = images.mosaic(W4, H4)
            if hashlib.sha256(mos.tobytes()).hexdigest() == gold["rgb_sha256"]:
                host.process(mos, quality=QUALITY, device=local_rank)
                env.sync()
                runs = []
                for _ in range(3):
                    tm = time.perf_counter()
                    jm, im = host.process(mos, quality=QUALITY, device=local_rank)
                    runs.append(time.perf_counter() - tm)
                tm = sorted(runs)[1]
                assert hashlib.sha256(jm).hexdigest() == gold["jpeg_sha256"], "mosaic output differs from the reference"
                other["mosaic_3840x2160_q95"] = {
                    "workload": "3840x2160 mosaic of the nine committed photographs (tests/images.mosaic: no "
                                "period, no RNG), --quality 95, one untimed encode, then the median of three timed ones",
                    "seconds": round(tm, 3), "value": round(W4 * H4 / 1e6 / tm, 3), "unit": "MPix/s",
                    "iterations": im["counters"].get("number of iterations"), "output_bytes": len(jm),
                    "output_sha256_matches_reference": True,
                    "reference_cpu_seconds": gold.get("reference_cpu_seconds"),
                    "host_timers_s": {k: round(v, 3) for k, v in im["timers"].items()
                                      if k in ("total", "phase_b_host", "compare", "block_search", "select_quant_matrix")}}
    c5 = c5png = None
    if not args.no_4k and not args.no_config5:   # every rank takes part
        c5 = config5_leg(env, host, images, args.images_per_gpu, args.in_flight, size5, quality)
        c5png = config5_leg(env, host, images, 1 if emu else args.images_per_gpu, args.in_flight, size5, quality,
                            from_png=True)
    traffic = load_traffic()

    if rank == 0:
        if not emu:
            sha = hashlib.sha256(jpg).hexdigest()
            assert sha == GOLDEN_SHA_1080P_Q95, f"output JPEG differs from the reference: {sha}"
            if jpg4k is not None:
                sha = hashlib.sha256(jpg4k).hexdigest()
                assert sha == GOLDEN_SHA_4K[95], f"3840x2160 output JPEG differs from the reference: {sha}"
        timer_keys = ("total", "phase_b_host", "compare", "block_search", "jpeg_write", "create+encode",
                      "select_quant_matrix", "pb_device_partitions", "pb_device_descents",
                      "pb_device_fetches", "pb_loop_codes")

        def leg(w, h, seconds, jpeg, inf, which):
            """value / ms_per_step / config of one timed size (both sizes are timed identically)."""
            return (round(world * args.steps * w * h / 1e6 / seconds, 4), round(seconds / args.steps * 1e3, 2),
                    {"workload": f"single {w}x{h} sRGB image, --quality {quality:g} (BASELINE {which}), whole "
                                 "guetzli::Process per step (host RGB in, JPEG bytes out), 1 image per GPU "
                                 f"per step; {args.steps} timed steps after {args.warmup} warm-up between "
                                 "barrier + synchronise, max over ranks",
                     "butteraugli_target": TARGET_Q95, "images_per_gpu": 1, "steps": args.steps,
                     "warmup": args.warmup, "output_bytes": len(jpeg),
                     "output_sha256_matches_reference": not emu,
                     "iterations": inf["counters"].get("number of iterations"),
                     # host threads of one encode: the search driver's own + the size model's code-refresh
                     # helpers (guetzli_amd/host/code_refresh.h; `pb_loop_codes` is the driver WAITING for them)
                     "host_code_refresh_threads": inf["counters"].get("phase B code refresh threads"),
                     "host_timers_s": {k: round(v, 3) for k, v in inf["timers"].items() if k in timer_keys}})

        def roof(w, h, ms_c, ach, key):
            t = traffic.get(key, {})
            return {"bound": "hbm", "kernel": CHAIN, "workload": f"{w}x{h}",
                    "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBPS, 4),
                    "traffic": None if traffic.get("stale") else t.get("traffic_bytes"),
                    "traffic_stale": bool(traffic.get("stale")), "traffic_head": traffic.get("head"),
                    "traffic_source": traffic.get("source"), "ms_per_compare": round(ms_c, 4),
                    "algorithmic_bytes_per_compare": ALGO_BYTES_PER_PX * w * h,
                    "valu": None if emu else valu_floor(key, ms_c),
                    "kernels": None if emu else (kernels or {}).get(key),
                    "kernels_source": None if emu or not kernels else
                                      {k: kernels.get(k) for k in ("source", "head", "stale") if k in kernels}}

        kernels = None if emu else load_kernel_roofline()
        v_small, ms_small, cfg_small = leg(W, H, dt, jpg, info, "configs[1]")
        roof_small = roof(W, H, ms, achieved, "1080p")
        if dt4k is not None:
            # the headline: BASELINE configs[2], north_star's target size and the rocprof config
            v_big, ms_big, cfg_big = leg(W4, H4, dt4k, jpg4k, info4k, "configs[2]")
            head, inf_head = (v_big, ms_big, cfg_big, roof(W4, H4, ms_4k, achieved_4k, "4k"), (W4, H4)), info4k
        else:
            head, inf_head = (v_small, ms_small, cfg_small, roof_small, (W, H)), info
        out = {
            "metric": "MPix/s encoded at --quality 95",
            "value": head[0],
            "value_workload": f"{head[4][0]}x{head[4][1]}",
            "unit": "MPix/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head[1],
            "higher_is_better": True, "scaling": "weak",
            # BASELINE.md section 2 measured this metric on this config (one thread of the survey container's CPU);
            # per GPU, so that the figure means the same at every N
            "vs_baseline": None if emu else round(head[0] / world / BASELINE_MD_MPIX_S[f"{head[4][0]}x{head[4][1]}"], 1),
            "vs_baseline_note": "value per GPU / BASELINE.md section 2's reference figure for this config (C3: 0.00746 "
                                "MPix/s, C2: 0.00770; one thread, Xeon @ 2.10 GHz); the same reference on one thread of "
                                "the bench box's kind of host: cpu_baseline.full_images_same_box_kind",
            "dtype": "f32+f64 (butteraugli), int32/int16 (DCT/quantize/entropy coding)",
            "data": f"synthetic (tests/golden/bees.png tiled to {head[4][0]}x{head[4][1]}, SURVEY 8d)",
            "config": head[2],
            "first_encode_s": round(first_encode_s, 3),
            "first_encode_4k_s": None if first_encode_4k_s is None else round(first_encode_4k_s, 3),
            "roofline": head[3],
            # phase A (not HBM-bound -- reported in evaluations, not bytes)
            "block_search": {"evaluations": inf_head["counters"].get("block search evaluations"),
                             "seconds": round(inf_head["timers"].get("block_search", 0.0), 4),
                             "evaluations_per_s": round(inf_head["counters"].get("block search evaluations", 0) /
                                                        max(inf_head["timers"].get("block_search", 0.0), 1e-9)),
                             "valu": block_search_counters(),
                             "note": "CompareBlock evaluations (one 8x8 IDCT + colour + opsin + FFT "
                                     "distance each) of gz_block_zeroing_orders on the headline image; "
                                     "`valu`: SQ counters of k_block_search<0> on the 1080p image from the "
                                     "committed --pmc pass (valu_active_per_wave_cycle x resident waves per "
                                     "SIMD = share of SIMD cycles issuing VALU)"},
            "host_timers_s": head[2]["host_timers_s"],
        }
        # the same two legs under names that do not depend on which of them is the headline
        out["value_1080p"] = v_small
        out["ms_per_step_1080p"] = ms_small
        out["step_ms_1080p"] = step_ms_1080p      # rank 0's individual timed steps
        out["step_ms_4k"] = step_ms_4k
        out["compares_4k"] = compares_4k          # (rank 0, warm-up + timed steps)
        if dt4k is not None:
            out["value_4k"] = head[0]
            out["ms_per_step_4k"] = head[1]
            out["config_1080p"] = cfg_small
            out["roofline_1080p"] = roof_small
            if kernels and not kernels.get("stale"):
                out["block_passes"] = kernels.get("block_passes")
        if small is not None:
            out["value_1mpix"] = small["value"]
            out["ms_per_step_1mpix"] = small["ms_per_step"]
            out["iteration_floor_us"] = small.get("iteration_floor_us")
            out["config_1mpix"] = small
        if c5 is not None:
            other = dict(other or {})
            other["config5_slice"] = c5
            other["config5_slice_from_png"] = c5png
            out["scale_value"] = c5["value"]
            out["scale_metric"] = ("MPix/s over BASELINE config 5's batch slice: 8 independent 3840x2160 "
                                   f"images per GPU, {args.in_flight} in flight per GPU")
        if other is not None:
            out["other_configs"] = other
        if batch is not None:
            out["batch_one_gpu"] = batch
        if not emu:
            cal = box_calibration()
            if cal:
                out["box"] = cal
        if world == 1 and not args.no_cpu_baseline and not emu:
            out["cpu_baseline"] = cpu_baseline()
            try:   # the same reference, one thread of this kind of host, on the headline image (committed record)
                same = out["cpu_baseline"]["full_images_same_box_kind"][f"{head[4][0]}x{head[4][1]}_q95"]
                out["vs_reference_same_box"] = round(head[0] / same["value"], 1)
            except (KeyError, TypeError):
                pass
        print(json.dumps(out), flush=True)
    env.finish()


if __name__ == "__main__":
    main()
