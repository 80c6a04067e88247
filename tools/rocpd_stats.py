#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (ROCm 7 default output of
`rocprofv3 --kernel-trace --stats`) into the per-kernel table `--stats` prints:
calls, total / average / min / max duration.  Usage: rocpd_stats.py results.db [out.csv]"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    ncol = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {ncol}, count(*), sum(end-start), avg(end-start), min(end-start), "
                       f"max(end-start) from kernels group by {ncol} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    lines = ["kernel,calls,total_ns,avg_ns,min_ns,max_ns,percent"]
    for n, c, t, a, mn, mx in rows:
        lines.append(f"\"{short(n)}\",{c},{t},{a:.0f},{mn},{mx},{100.0 * t / total:.2f}")
    text = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    sys.stdout.write(text)


if __name__ == "__main__":
    main()
