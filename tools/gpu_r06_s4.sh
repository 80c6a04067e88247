#!/bin/bash
# Round 6, session 4: the refactored host driver on the GPU suite; helper threads / images in flight under the
# box's CPU quota (16 CPUs' worth behind a 256-CPU affinity mask), batch mode.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06d; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
( timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 ) | tee $O/gputests.log
{
for rep in 1 2; do
  for cfg in "GZ_NONE=1" "GZ_CODE_THREADS=3" "GZ_CODE_THREADS=1" "GZ_CODE_THREADS=0" "GZ_HOST_THREADS=4"; do
    echo "== $cfg"
    env $cfg python tools/batch_time.py 3840 2160 8 4 2
    env $cfg python tools/batch_time.py 1920 1080 16 4 2
    env $cfg python tools/batch_time.py 1024 1024 64 4 2
    env $cfg python tools/batch_time.py 1024 1024 64 6 2
  done
done
echo "== single encodes"
for cfg in "GZ_NONE=1" "GZ_CODE_THREADS=3"; do echo "== $cfg"; env $cfg python tools/encode_time.py 3840 2160 95 4 2>&1 | tail -2; env $cfg python tools/encode_time.py 1024 1024 95 6 2>&1 | tail -2; done
} 2>&1 | tee $O/threads.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err
