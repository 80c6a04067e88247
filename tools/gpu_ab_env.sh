#!/bin/bash
# A/B of environment switches with the chain's per-kernel durations: for each configuration the
# three-stream chain by HIP events (4K, 1080p; two repetitions, interleaved), one rocprofv3
# --kernel-trace --stats pass of the chain serialised on one stream at 4K, optionally GPU tests.
# Usage: gpurun -- 'bash tools/gpu_ab_env.sh TAG [--tests "pytest -k expr"] "-" "A=1" "A=1 B=2" ...'
set -u
export TMPDIR=/tmp
TAG=$1; shift
TESTS=""; [ "${1:-}" = "--tests" ] && { TESTS=$2; shift 2; }
R=$GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2; do
  for cfg in "$@"; do
    e=$cfg; [ "$cfg" = "-" ] && e="GZ_NONE=1"
    echo "== $cfg"
    env $e python tools/run_compare.py 3840 2160 100
    env $e python tools/run_compare.py 1920 1080 200
  done
done
} 2>&1 | tee $O/chain.log
i=0
for cfg in "$@"; do
  e=$cfg; [ "$cfg" = "-" ] && e="GZ_NONE=1"
  d=$O/trace_$i
  ( cd /tmp && env $e GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -- python $R/tools/run_compare.py 3840 2160 20 ) > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; }
  if [ -n "$TESTS" ]; then echo "== tests under $cfg"; env $e timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "$TESTS" 2>&1 | tail -3; fi
  i=$((i+1))
done 2>&1 | tee $O/tests.log
python3 - $O "$@" <<'PY'
import csv, re, sys, os
O, cfgs = sys.argv[1], sys.argv[2:]
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r'\(.*$', '', re.sub(r'^void ', '', r['Name'])).replace('gz::', '')
        d[n] = (int(r['Calls']), float(r['AverageNs']) / 1e3)
    return d
tabs = [load(os.path.join(O, f"trace_{i}.csv")) for i in range(len(cfgs)) if os.path.exists(os.path.join(O, f"trace_{i}.csv"))]
names = sorted({k for t in tabs for k in t if t[k][0] >= 20}, key=lambda k: -max(t.get(k, (0, 0))[1] for t in tabs))
print("kernel (us, chain serialised on one stream, 4K)".ljust(62) + "".join(c[:18].rjust(20) for c in cfgs))
for k in names[:20]:
    print(k[:60].ljust(62) + "".join(f"{t[k][1]:20.1f}" if k in t else " " * 20 for t in tabs))
print("sum".ljust(62) + "".join(f"{sum(v[1] for k, v in t.items() if v[0] >= 20):20.1f}" for t in tabs))
PY
