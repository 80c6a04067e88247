#!/bin/bash
# Where the host's time goes at the end of a phase-B iteration (compare_begin / jpeg_scan_begin /
# order-ahead begin / jpeg_scan_end / compare_end), with and without the helper-thread head.
# Usage: tools/gpurun_head.sh --timeout 600 -- 'bash tools/gpu_r3_tail.sh [tag] ["ENV=.."...]'
set -u
export TMPDIR=/tmp
TAG=${1:-tail}; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
nproc | tee $O/tail.log
for cfg in "$@"; do
  for sz in "1920 1080 95 6" "3840 2160 95 3"; do
    echo "== $cfg $sz"; env $cfg python tools/encode_time.py $sz | sed 's/; iters.*//' | tr ',' '\n' | grep -v "sha" | paste -sd' ' | fold -w 200
  done
done 2>&1 | tee -a $O/tail.log
