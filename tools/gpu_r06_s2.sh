#!/bin/bash
# Round 6, session 2: co-residency knobs judged in batch mode and on the three-stream chain:
# side-branch kernels in Malta-sized forms (GZ_SIDE_SMALL), Malta bounded to 7 / 6 / 4 workgroups per CU by unused
# dynamic LDS (GZ_MALTA_PAD), 16-row tiles at 4K; then the whole bench line with the round's new legs.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06b; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2; do
  for cfg in "GZ_NONE=1" "GZ_SIDE_SMALL=1" "GZ_MALTA_PAD=3400" "GZ_MALTA_PAD=7400" "GZ_MALTA_PAD=13000" \
             "GZ_SIDE_SMALL=1 GZ_MALTA_PAD=7400" "GZ_TILE_ROWS=16" "GZ_TILE_ROWS=16 GZ_MALTA_PAD=7400" "GZ_TILE_ROWS=16 GZ_SIDE_SMALL=1 GZ_MALTA_PAD=3400"; do
    echo "== $cfg"
    env $cfg python tools/run_compare.py 3840 2160 100
    env $cfg python tools/run_compare.py 1920 1080 200
    env $cfg python tools/batch_time.py 3840 2160 8 4 2
    env $cfg python tools/batch_time.py 1920 1080 16 4 2
  done
done
} 2>&1 | tee $O/coresidency.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
