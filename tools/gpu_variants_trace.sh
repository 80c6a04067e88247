#!/bin/bash
# A/B of BUILD-TIME variants (guetzli_amd/variants/<name>.so) with the chain's per-kernel durations:
# for each variant the three-stream chain by HIP events (4K, 1080p; two repetitions, interleaved) and
# one rocprofv3 --kernel-trace --stats pass of the chain serialised on one stream at 4K.
# Usage: gpurun -- 'bash tools/gpu_variants_trace.sh TAG name1 name2 ...'  ("base" = the built library)
set -u
export TMPDIR=/tmp
TAG=$1; shift
R=$GRAFT_REPO_ROOT
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
use() { if [ "$1" = base ]; then cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so; else cp guetzli_amd/variants/$1.so guetzli_amd/libguetzli_amd.so; fi; }
{
for rep in 1 2; do
  for v in "$@"; do
    use $v
    echo "== $v"
    python tools/run_compare.py 3840 2160 100
    python tools/run_compare.py 1920 1080 200
  done
done
} 2>&1 | tee $O/chain.log
for v in "$@"; do
  use $v
  d=$O/trace_$v
  ( cd /tmp && GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -- python $R/tools/run_compare.py 3840 2160 20 ) > $d.log 2>&1
  f=$(find $d -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; echo "== $v"; cut -d, -f1-4 $d.csv | sed 's/gz:://g; s/void //' | awk -F'"' '{printf "%-64s %s\n", substr($2,1,64), $NF}' | head -20; }
done 2>&1 | tee $O/trace.log
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
