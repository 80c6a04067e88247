#!/bin/bash
# Round 6, session 6: batch mode, one stream per image against three, by image size and images in flight.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06f; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2 3; do
  for cfg in "GZ_NONE=1" "GZ_SINGLE_STREAM=1"; do
    echo "== $cfg"
    for wk in 4 6; do
      env $cfg python tools/batch_time.py 512 512 128 $wk 1
      env $cfg python tools/batch_time.py 1024 1024 64 $wk 1
      env $cfg python tools/batch_time.py 1920 1080 16 $wk 1
      env $cfg python tools/batch_time.py 2560 1440 12 $wk 1
      env $cfg python tools/batch_time.py 3840 2160 8 $wk 1
    done
  done
done
} 2>&1 | tee $O/batch_streams.log
