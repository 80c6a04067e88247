#!/bin/bash
# A/B of environment knobs on the MI355X: the Compare chain by HIP events (three streams) at 1080p
# and 4K, whole encodes at both sizes (tools/encode_time.py: median of the runs after the first),
# two repetitions, optionally the per-kernel statistics of the chain serialised on one stream.
# Usage: gpurun --timeout 900 -- 'bash tools/gpu_ab.sh TAG [--trace] "A=0" "A=1 B=2" ...'
#        (a configuration may hold several assignments; "-" = no assignment)
set -u
export TMPDIR=/tmp
TAG=${1:-ab}; shift
TRACE=0; [ "${1:-}" = "--trace" ] && { TRACE=1; shift; }
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
tools/ubench/bw 2>/dev/null | head -3 | tee $O/bw.log
{
for rep in 1 2; do
  for cfg in "$@"; do
    e=$cfg; [ "$cfg" = "-" ] && e="GZ_NONE=1"
    echo "== $cfg"
    env $e python tools/run_compare.py 1920 1080 100
    env $e python tools/run_compare.py 3840 2160 40
    env $e python tools/encode_time.py 1920 1080 95 x 4 | head -1 | cut -c1-120
    env $e python tools/encode_time.py 3840 2160 95 x 4 | head -1 | cut -c1-120
  done
done
} 2>&1 | tee $O/ab.log
if [ $TRACE = 1 ]; then
  i=0
  for cfg in "$@"; do
    e=$cfg; [ "$cfg" = "-" ] && e="GZ_NONE=1"
    for sz in "3840 2160 20" "1920 1080 40"; do
      d=$O/trace_${i}_$(echo $sz | cut -d' ' -f1)
      ( cd /tmp && env $e GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/tools/run_compare.py $sz ) > $d.log 2>&1
      f=$(find $d -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; echo "== $cfg $sz"; cut -d, -f1-4 $d.csv | sed 's/gz:://; s/void //' | cut -c1-70,150- | head -24; }
    done
    i=$((i+1))
  done 2>&1 | tee $O/trace.log
fi
