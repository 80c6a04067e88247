#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-cab}; mkdir -p $O
{ for sz in "3840 2160 30" "1920 1080 60" "1280 720 80" "444 258 100"; do
  for v in "0 0" "1 1" "0 0" "1 1"; do set -- $v; echo -n "COMPACT_V=$1 2D=$2  "; GZ_COMPACT_BLUR_V=$1 GZ_COMPACT_BLUR2D=$2 python tools/run_compare.py $sz; done; done; } 2>&1 | tee $O/compact_ab.log
