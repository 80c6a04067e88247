#!/bin/bash
# Round 6, session 11: what runs beside what in batch mode (kernel trace of 4 x 4K, four in flight), with one
# stream per image (the default) and with three.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06k; mkdir -p $O
R=$GRAFT_REPO_ROOT
cp .gpurun_head $O/head.txt 2>/dev/null || true
for mode in default three; do
  e="GZ_NONE=1"; [ $mode = three ] && e="GZ_SINGLE_STREAM=0"
  ( cd /tmp && env $e timeout 400 rocprofv3 --kernel-trace --output-format csv -d $R/$O/trace_$mode -- python $R/tools/batch_time.py 3840 2160 4 4 1 ) > $O/trace_$mode.log 2>&1
  f=$(find $O/trace_$mode -name "*kernel_trace.csv" | head -1)
  [ -n "$f" ] && python tools/batch_overlap.py $f profiles/r06_compare_4k_kernel_stats_single_stream.csv > $O/batch_overlap_$mode.txt 2>&1
  rm -rf $O/trace_$mode
  tail -2 $O/trace_$mode.log
done
cat $O/batch_overlap_default.txt
