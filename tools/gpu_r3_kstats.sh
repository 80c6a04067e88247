#!/bin/bash
# rocprofv3 kernel statistics of one command, filtered: gpu_r3_kstats.sh TAG PATTERN -- cmd...
set -u
export TMPDIR=/tmp
TAG=$1; PAT=$2; shift 3
O=gpurun_out/$TAG; mkdir -p $O
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace -- "$@" ) > $O/trace.log 2>&1
f=$(find $O/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && { cp $f $O/kernel_stats.csv; grep -E "$PAT" $f | sed 's/gz:://g' | awk -F'",' '{print substr($1,2,50) "  " $2}'; }
rm -rf $O/trace
