#!/bin/bash
# Round 6, session 9: when Malta starts relative to the side branches (GZ_MALTA_ORDER), lone context, three streams.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06i; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2 3; do
  for cfg in "GZ_MALTA_ORDER=0" "GZ_MALTA_ORDER=1" "GZ_MALTA_ORDER=2" "GZ_MALTA_ORDER=3"; do
    echo "== $cfg"
    env $cfg python tools/run_compare.py 3840 2160 200
    env $cfg python tools/run_compare.py 1920 1080 400
    env $cfg python tools/run_compare.py 1024 1024 400
  done
done
for cfg in "GZ_MALTA_ORDER=0" "GZ_MALTA_ORDER=1" "GZ_MALTA_ORDER=2" "GZ_MALTA_ORDER=3"; do
  echo "== encodes $cfg"
  env $cfg python tools/encode_time.py 3840 2160 95 5 | head -1 | cut -c1-120
  env $cfg python tools/encode_time.py 1920 1080 95 7 | head -1 | cut -c1-120
done
( timeout 600 env GZ_MALTA_ORDER=1 python -m pytest tests/test_gpu_parity.py -x -q -k "compare or stages or config" 2>&1 | tail -2 )
( timeout 600 env GZ_MALTA_ORDER=2 python -m pytest tests/test_gpu_parity.py -x -q -k "compare_bees or config" 2>&1 | tail -2 )
( timeout 600 env GZ_MALTA_ORDER=3 python -m pytest tests/test_gpu_parity.py -x -q -k "compare_bees or config" 2>&1 | tail -2 )
} 2>&1 | tee $O/malta_order.log
