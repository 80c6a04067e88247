#!/bin/bash
# Merged timeline (HIP API calls on the host, kernels and copies on the device) of phase-B
# iterations of one encode.  Usage: gpu_trace_full.sh TAG [W H]   (default 1920 1080)
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-trf}; mkdir -p $O
W=${2:-1920}; H=${3:-1080}
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- python $GRAFT_REPO_ROOT/tools/encode_time.py $W $H 95 x 2 ) > $O/trace.log 2>&1
tail -2 $O/trace.log
python3 - $O/trace <<'PY' | tee $O/timeline_full.txt
import csv, sys, glob, os
d = sys.argv[1]
def load(pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return list(csv.DictReader(open(fs[0]))) if fs else []
k = load("*kernel_trace.csv"); m = load("*memory_copy_trace.csv"); a = load("*hip_api_trace.csv")
ev = []
for r in k: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s " % r.get("Queue_Id", "") + r["Kernel_Name"].replace("gz::", "").split("(")[0][:48]))
for r in m: ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C " + r.get("Direction", "") + " " + r.get("Bytes", r.get("Size", ""))))
for r in a:
    n = r.get("Function", r.get("Name", ""))
    if n in ("hipGetDevice", "hipSetDevice", "hipGetLastError", "hipPeekAtLastError", "__hipPushCallConfiguration", "__hipPopCallConfiguration"): continue
    ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "A " + n))
ev.sort()
rec = [i for i, e in enumerate(ev) if "PostOpsin" in e[2] and e[2].startswith("K ")]   # (one per Compare)
i0 = rec[len(rec) // 2 + 20]; i1 = rec[len(rec) // 2 + 22]
# start a little before the reconstruct kernel (its launch call)
t0 = ev[i0][0]
for s, e, n in ev[max(0, i0 - 40):i1]:
    if s < t0 - 200000: continue
    print("%10.1f us  dur %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, n))
PY
rm -rf $O/trace
