#!/bin/bash
# Round 6, session 5: small images -- is the three-stream chain (fork / join events: host time per Compare) still
# worth it where kernels take microseconds?  Whole encodes and chains at 64^2 .. 1080p, one stream against three.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06e; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2; do
  for cfg in "GZ_NONE=1" "GZ_SINGLE_STREAM=1"; do
    echo "== $cfg"
    for sz in "64 64" "256 256" "512 512" "1024 1024" "1920 1080"; do
      env $cfg python tools/encode_time.py $sz 95 8 | head -1 | cut -c1-110
      env $cfg python tools/run_compare.py $sz 300
    done
    env $cfg python tools/batch_time.py 1024 1024 64 4 2
    env $cfg python tools/batch_time.py 512 512 128 4 2
  done
done
} 2>&1 | tee $O/small.log
