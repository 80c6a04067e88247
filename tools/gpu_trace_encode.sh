#!/bin/bash
# Kernel timeline of one 1080p encode (second of two): where the GPU idles between kernels.
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-tr}; mkdir -p $O
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/encode_time.py 1920 1080 95 x 2 ) > $O/trace.log 2>&1
tail -3 $O/trace.log
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python - $f <<'PY' | tee $O/timeline.txt
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = int(rows[0]["Start_Timestamp"])
# second encode = second half of the k_reconstruct launches
rec = [i for i, r in enumerate(rows) if "k_reconstruct" in r["Kernel_Name"]]
half = rec[len(rec) // 2 + 20]   # an iteration well inside phase B of the second encode
end = rec[len(rec) // 2 + 23]
prev_end = int(rows[half - 1]["End_Timestamp"])
for r in rows[half:end]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("gz::", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:12.1f} us  +gap {(s - prev_end) / 1e3:7.1f}  dur {(e - s) / 1e3:7.1f}  q{r.get('Queue_Id','')} {name}")
    prev_end = max(prev_end, e)
PY
