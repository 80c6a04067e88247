#!/bin/bash
# Round 6, session 23: the opsin blur ahead of the serial steps (gz_config.opsin_ahead) -- GPU suite, A/B of whole
# encodes, a timeline.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06aa; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -3 | tee $O/tests.log
{
for rep in 1 2 3; do
  for e in GZ_OPSIN_AHEAD=1 GZ_OPSIN_AHEAD=0; do
    echo "== $e"
    env $e python tools/ab_old_time.py . 3840 2160 7
    env $e python tools/ab_old_time.py . 1920 1080 9
    env $e python tools/ab_old_time.py . 1024 1024 9
    env $e python tools/ab_old_time.py . 512 512 9
  done
done
for e in GZ_OPSIN_AHEAD=1 GZ_OPSIN_AHEAD=0; do
  echo "== $e"
  env $e python tools/batch_time.py 3840 2160 8 4 2
  env $e python tools/batch_time.py 1920 1080 16 4 2
done
} 2>&1 | cut -c1-110 | tee $O/ab.log
bash tools/gpu_trace_full.sh r06aa 3840 2160 > /dev/null
bash tools/gpu_trace_full.sh r06aa1080 1920 1080 > /dev/null
