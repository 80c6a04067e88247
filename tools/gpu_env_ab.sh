#!/bin/bash
# A/B of an environment knob of the device library: chain parity tests once, then chain time at
# 1080p and 4K per value, then single-stream kernel statistics for the LAST value.
# Usage: gpu_env_ab.sh TAG VAR=a,b
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
KV=$2; VAR=${KV%%=*}; VALS=${KV#*=}
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or stages or compare or frame420 or paired or full_size" 2>&1 | tail -3 ) | tee $O/pytest.log
{
for rep in 1 2; do for v in ${VALS//,/ }; do
  echo "== $VAR=$v"; env $VAR=$v python tools/run_compare.py 1920 1080 100; env $VAR=$v python tools/run_compare.py 3840 2160 40
done; done
for v in ${VALS//,/ }; do echo "== encode $VAR=$v"; env $VAR=$v python tools/encode_time.py 1920 1080 95 5 | head -1 | cut -c1-120; done
} 2>&1 | tee $O/ab.log
for sz in "3840 2160 20 4k" "1920 1080 40 1080"; do set -- $sz
  ( cd /tmp && GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$4 -- python $GRAFT_REPO_ROOT/tools/run_compare.py $1 $2 $3 ) > $O/trace$4.log 2>&1
  f=$(find $O/trace$4 -name "*kernel_stats.csv" | head -1)
  echo "== $4 (single stream)"; python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-60s %5s %9.1f'%(r['Name'].replace('gz::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
  cp $f $O/kernel_stats_$4.csv
  rm -rf $O/trace$4
done
