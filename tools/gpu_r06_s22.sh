#!/bin/bash
# Round 6, session 22: k_apply_steps_hist with the "precious" test from a wavefront sum; GPU suite; encode times;
# a timeline of the 4K iteration.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06y; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -3 | tee $O/tests.log
{
for rep in 1 2 3; do
  python tools/encode_time.py 3840 2160 95 8 | head -1 | cut -c1-150
  python tools/encode_time.py 1920 1080 95 10 | head -1 | cut -c1-150
  python tools/encode_time.py 1024 1024 95 10 | head -1 | cut -c1-150
done
python tools/batch_time.py 3840 2160 8 4 2
python tools/batch_time.py 1920 1080 16 4 2
python tools/batch_time.py 1024 1024 64 6 2
} 2>&1 | tee $O/ab.log
bash tools/gpu_trace_full.sh r06y 3840 2160 > /dev/null
bash tools/gpu_trace_full.sh r06y1080 1920 1080 > /dev/null
