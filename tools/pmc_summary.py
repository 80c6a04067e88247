#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 --pmc CSV output (counter_collection.csv files under DIR).
Usage: pmc_summary.py DIR [DIR...]  -> CSV on stdout: kernel,counter,calls,avg_value"""
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: [0, 0.0])
for d in sys.argv[1:]:
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            name = re.sub(r"\(.*$", "", re.sub(r"^void ", "", row["Kernel_Name"]))
            k = (name, row["Counter_Name"])
            acc[k][0] += 1
            acc[k][1] += float(row["Counter_Value"])
print("kernel,counter,calls,avg_value")
for (name, ctr), (n, s) in sorted(acc.items()):
    print(f"\"{name}\",{ctr},{n},{s / n:.1f}")
