#!/bin/bash
# One GPU session (run through gpurun): the -m gpu suite, the default bench, chain timings, and
# the rocprofv3 kernel statistics of the bench command (copied to profiles/ by hand afterwards).
# Usage: gpurun --timeout 1800 -- 'bash tools/gpu_session.sh [tag]'
set -u
export TMPDIR=/tmp
TAG=${1:-s}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest.log
python bench.py 2>$O/bench.err | tee $O/bench.json
python tools/run_compare.py 3840 2160 30 | tee $O/compare_4k.log
python tools/run_compare.py 1920 1080 60 | tee $O/compare_1080.log
python tools/encode_time.py 1920 1080 95 5 | tee $O/encode_1080p.log
python tools/encode_time.py 1920 1080 95 force_420 5 | tee $O/encode_1080p_420.log
python tools/encode_time.py 3840 2160 95 3 | tee $O/encode_4k.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -- python $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-4k --no-config5 --batch-images 0 ) > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv
tail -2 $O/prof.log
