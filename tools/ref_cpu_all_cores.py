#!/usr/bin/env python3
"""The UNMODIFIED reference on ALL host cores at once, the way the reference's own batch form runs it
(tests/golden_test.sh:24-26: one guetzli process per image under `xargs -P`): one worker PROCESS per
allowed CPU, each encoding one W x H sample (the bench image's top-left 640x360 by default) at
--quality 95 from a common start; aggregate MPix/s = workers x pixels / the time until the LAST one is
done.  bench.py quotes the record as cpu_baseline.all_cores beside the per-GPU batch numbers (a bounded
sample: ~15-25 s whatever the core count).  Started as its own process so that the workers are forked
from an interpreter that never loaded HIP or PyTorch.
Usage: ref_cpu_all_cores.py [W H [WORKERS]]  -> one JSON line"""
import hashlib, json, multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

TARGET_Q95 = 0.971769


def worker(k, cpu, w, h, start, q):
    try:
        os.sched_setaffinity(0, [cpu])
    except OSError:
        pass
    import images
    from checkers import ref
    rgb = images.tiled(w, h)
    start.wait()
    t0 = time.perf_counter()
    jpg, _ = ref.process(rgb, TARGET_Q95)
    q.put((k, time.perf_counter() - t0, hashlib.sha256(jpg).hexdigest()))


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 360
    cpus = sorted(os.sched_getaffinity(0))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else len(cpus)
    ctx = mp.get_context("fork")
    start = ctx.Barrier(n + 1)
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(k, cpus[k % len(cpus)], w, h, start, q)) for k in range(n)]
    for p in procs:
        p.start()
    start.wait()
    t0 = time.perf_counter()
    got = [q.get() for _ in range(n)]
    dt = time.perf_counter() - t0
    for p in procs:
        p.join()
    model = next((l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")), "?")
    secs = sorted(g[1] for g in got)
    print(json.dumps({
        "value": round(n * w * h / 1e6 / dt, 5), "unit": "MPix/s", "cores": n, "kind": "reference",
        "seconds": round(dt, 2), "seconds_fastest_worker": round(secs[0], 2), "seconds_slowest_worker": round(secs[-1], 2),
        "host_cpu": model, "host_cpus_present": os.cpu_count(),
        "distinct_outputs": len({g[2] for g in got}),
        "sample": f"{n} processes (one per allowed logical CPU, pinned), each the unmodified reference guetzli::Process on the "
                  f"top-left {w}x{h} of the bench image at --quality 95, started together (the form of "
                  "tests/golden_test.sh:24-26); value = processes x pixels / time until the last one finished"}))


if __name__ == "__main__":
    main()
