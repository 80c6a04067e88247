#!/usr/bin/env python3
"""What runs beside what in batch mode, from a rocprofv3 --kernel-trace CSV of tools/batch_time.py (VERDICT r5 item 1:
"which kernels' waves are resident together ... what else blocks the overlap").

  * share of the traced span with 0 / 1 / 2 / 3 / 4+ kernels in flight (device idle, one kernel alone, ...);
  * per kernel: launches, time in the batch, mean duration in the batch against its duration alone (the committed
    single-stream statistics of the same kernel sources) = the slow-down it pays for its company;
  * the pairs of kernels that overlap most (share of the span in which both are in flight).

Usage: batch_overlap.py kernel_trace.csv [solo_kernel_stats.csv]"""
import collections
import csv
import re
import sys


def short(n):
    return re.sub(r"\(.*$", "", n.replace("void ", "")).replace("gz::", "")


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in rows)
    solo = {}
    if len(sys.argv) > 2:
        for r in csv.DictReader(l for l in open(sys.argv[2]) if not l.startswith("#")):
            solo[short(r["Name"])] = float(r["AverageNs"])
    # the steady part: from the first to the last k_reconstruct (context creation and the first encode's set-up out)
    rec = [e for e in ev if e[2].startswith("k_reconstruct")]
    t0, t1 = rec[len(rec) // 10][0], rec[-1][1]
    ev = [e for e in ev if e[0] >= t0 and e[1] <= t1]
    span = t1 - t0
    points = []
    for i, (s, e, _) in enumerate(ev):
        points.append((s, 1, i))
        points.append((e, -1, i))
    points.sort()
    depth_time = collections.Counter()
    pair_time = collections.Counter()
    active = set()
    last = t0
    for t, d, i in points:
        dt = t - last
        if dt > 0:
            depth_time[min(len(active), 4)] += dt
            if 2 <= len(active) <= 6:
                names = sorted({ev[j][2] for j in active})
                for a in range(len(names)):
                    for b in range(a + 1, len(names)):
                        pair_time[(names[a], names[b])] += dt
        last = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    print(f"# {len(ev)} kernels over {span / 1e6:.1f} ms of the batch (first tenth of the evaluations left out)")
    print("kernels in flight   share of the span")
    for k in range(5):
        print(f"  {k}{'+' if k == 4 else ' '}                  {depth_time[k] / span:6.3f}")
    per = collections.defaultdict(lambda: [0, 0])
    for s, e, n in ev:
        per[n][0] += 1
        per[n][1] += e - s
    print("\nkernel                                                        launches  share of the span  mean us in the batch  alone us  ratio")
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:22]:
        a = solo.get(n)
        print(f"{n[:60]:60s} {c:8d} {t / span:12.3f} {t / c / 1e3:18.1f} "
              f"{(a / 1e3 if a else float('nan')):12.1f} {(t / c / a if a else float('nan')):6.2f}")
    busy = sum(t for _, t in per.values())
    print(f"\nsum of the kernels' durations / span = {busy / span:.2f} (kernel-seconds per second: > 1 means overlap)")
    print("\npairs in flight together (share of the span)")
    for (a, b), t in pair_time.most_common(12):
        print(f"  {t / span:6.3f}  {a[:44]:44s} + {b[:44]}")


if __name__ == "__main__":
    main()
