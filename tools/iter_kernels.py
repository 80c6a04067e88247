#!/usr/bin/env python3
"""Kernel timeline of ONE phase-B iteration of an encode from a rocprofv3 --kernel-trace CSV: the launches from
the N-th k_reconstruct to the next one (start relative to that kernel, duration, name).
Usage: iter_kernels.py kernel_trace.csv N"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2])
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_reconstruct" in r["Kernel_Name"]]
a, b = idx[n], idx[n + 1]
t0 = int(rows[a]["Start_Timestamp"])
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("gz::", "").replace("void ", "")
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  {name[:70]}")
