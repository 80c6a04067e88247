#!/bin/bash
# Generic A/B of the Compare chain under environment knobs: chain time by HIP events (3 streams) at
# 1080p and 4K, two repetitions, then per-kernel rocprofv3 statistics (single stream) per config.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_r3_ab.sh TAG "A=0" "A=1" ...'  (a config may hold several assignments)
set -u
export TMPDIR=/tmp
TAG=${1:-ab}; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
tools/ubench/bw 2>/dev/null | head -3 | tee $O/bw.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "malta or compare_bees or blur_code or stages" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
{
for rep in 1 2; do
  for cfg in "$@"; do
    echo "== $cfg"; env $cfg python tools/run_compare.py 1920 1080 100; env $cfg python tools/run_compare.py 3840 2160 40
  done
done
} 2>&1 | tee $O/chain_ab.log
i=0
for cfg in "$@"; do
  for sz in "3840 2160 20" "1920 1080 40"; do
    d=$O/trace_${i}_$(echo $sz | cut -d' ' -f1)
    ( cd /tmp && env $cfg GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/tools/run_compare.py $sz ) > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; echo "== $cfg $sz"; grep -i "malta" $d.csv | cut -d, -f1-4 | cut -c1-60,120-; }
  done
  i=$((i+1))
done
