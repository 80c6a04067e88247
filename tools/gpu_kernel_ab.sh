#!/bin/bash
# A/B of one chain kernel under environment knobs: per-kernel time with the chain serialised
# (rocprofv3 --kernel-trace --stats, GZ_SINGLE_STREAM=1), the chain by HIP events on three streams,
# whole encodes.  Usage: gpurun ... -- 'bash tools/gpu_kernel_ab.sh TAG KERNEL_SUBSTRING "A=0" "A=1" ...'
set -u
export TMPDIR=/tmp
TAG=${1:-kab}; PAT=${2:-k_malta}; shift 2
CFGS=("$@")
R=$GRAFT_REPO_ROOT; O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for cfg in "${CFGS[@]}"; do
  for sz in "3840 2160 10" "1920 1080 20"; do set -- $sz
    d=$O/t
    ( cd /tmp && env GZ_SINGLE_STREAM=1 $cfg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -- python $R/tools/run_compare.py $1 $2 $3 ) > $d.log 2>&1
    python3 - $d "$cfg" $1 "$PAT" <<'PY'
import csv, glob, os, sys
d, cfg, w, pat = sys.argv[1:5]
for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    if not any("k_combine" in r["Name"] for r in rows): continue
    tot = sum(float(r["TotalDurationNs"]) for r in rows if "rocclr" not in r["Name"] and not any(x in r["Name"] for x in ("k_encode_rgb", "k_linear_from", "k_quantize")))
    n = [int(r["Calls"]) for r in rows if "k_combine" in r["Name"]][0]
    for r in rows:
        if pat in r["Name"]:
            print(f"{cfg:24s} w={w} {r['Name'].split('(')[0].replace('void gz::',''):24s} avg {float(r['AverageNs'])/1e3:7.1f} us   chain kernels' sum {tot/n/1e3:7.1f} us")
PY
    rm -rf $d
  done
done
for rep in 1 2; do for cfg in "${CFGS[@]}"; do echo "== $cfg"; env $cfg python tools/run_compare.py 3840 2160 40; env $cfg python tools/run_compare.py 1920 1080 100; done; done
for rep in 1 2; do for cfg in "${CFGS[@]}"; do echo "== $cfg"; env $cfg python tools/encode_time.py 1920 1080 95 6 | head -1 | cut -c1-130; env $cfg python tools/encode_time.py 3840 2160 95 3 | head -1 | cut -c1-130; done; done
} 2>&1 | tee $O/ab.log
