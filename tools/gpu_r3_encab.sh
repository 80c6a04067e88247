#!/bin/bash
# Generic A/B of whole encodes under environment knobs: 1080p (6 runs, median of 5) and 4K (3 runs),
# two repetitions.  Usage: gpurun --timeout 900 -- 'bash tools/gpu_r3_encab.sh TAG "A=0" "A=1" ...'
set -u
export TMPDIR=/tmp
TAG=${1:-eab}; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2; do
  for cfg in "$@"; do
    echo "== $cfg"; env $cfg python tools/encode_time.py 1920 1080 95 6; env $cfg python tools/encode_time.py 3840 2160 95 3
  done
done
} 2>&1 | tee $O/encode_ab.log | cut -c1-150
