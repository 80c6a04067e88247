#!/bin/bash
# Kernel time of k_block_search per build-time variant (rocprofv3 kernel trace of tools/run_search.py).
# Usage: gpurun -- 'bash tools/gpu_search_trace.sh TAG name1 name2 ...'
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
R=$(pwd); export TMPDIR=/tmp
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
{
for v in "$@"; do
  cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd.so
  for sz in "1920 1080" "3840 2160"; do
    d=$O/tr_$v
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -- python $R/tools/run_search.py $sz ${SEARCH_ARGS:-} ) > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    echo "== $v $sz: $(grep -i 'k_block_search' $f | cut -d, -f2-4 | tr '\n' ' ')"
    rm -rf $d
  done
done
} 2>&1 | tee $O/trace.log
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
