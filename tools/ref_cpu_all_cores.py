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


def cpu_quota():
    """CPUs' worth of time the container may use (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited: a
    box may show 256 logical CPUs in its affinity mask and still be throttled to a fraction of them."""
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        return None if quota == "max" else float(quota) / float(period)
    except (OSError, ValueError):
        pass
    try:
        q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        return None if q <= 0 else q / p
    except (OSError, ValueError):
        return None


def worker(k, cpu, rgb, ref, start, q):
    if cpu is not None:
        try:
            os.sched_setaffinity(0, [cpu])
        except OSError:
            pass
    start.wait()
    t0 = time.perf_counter()
    jpg, _ = ref.process(rgb, TARGET_Q95)
    q.put((k, time.perf_counter() - t0, hashlib.sha256(jpg).hexdigest()))


def main():
    w = int(sys.argv[1]) if len(sys.argv) > 1 else 640
    h = int(sys.argv[2]) if len(sys.argv) > 2 else 360
    cpus = sorted(os.sched_getaffinity(0))
    quota = cpu_quota()
    usable = len(cpus) if quota is None else max(1, min(len(cpus), int(quota)))
    n = int(sys.argv[3]) if len(sys.argv) > 3 else usable
    pin = quota is None or quota >= len(cpus)     # under a quota the scheduler places the workers
    import images                                 # (loaded once, before the fork: 256 interpreters importing numpy
    from checkers import ref                      # at the same time cost more than the sample itself)
    rgb = images.tiled(w, h)
    ctx = mp.get_context("fork")
    start = ctx.Barrier(n + 1)
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(k, cpus[k % len(cpus)] if pin else None, rgb, ref, start, q))
             for k in range(n)]
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
        "host_cpu": model, "host_cpus_present": os.cpu_count(), "host_cpus_allowed": len(cpus),
        "cgroup_cpu_quota": quota, "pinned": pin,
        "distinct_outputs": len({g[2] for g in got}),
        "sample": f"{n} processes (one per usable logical CPU{', pinned' if pin else ''}), each the unmodified reference guetzli::Process on the "
                  f"top-left {w}x{h} of the bench image at --quality 95, started together (the form of "
                  "tests/golden_test.sh:24-26); value = processes x pixels / time until the last one finished"}))


if __name__ == "__main__":
    main()
