#!/bin/bash
# One GPU session (run through gpurun): the -m gpu suite, a short bench, kernel statistics.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tag]'
set -u
export TMPDIR=/tmp
TAG=${1:-s}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest.log
python bench.py --steps 10 --warmup 3 2>$O/bench.err | tee $O/bench.json
python tools/encode_time.py 1920 1080 95 force_420 2 2>&1 | tee $O/encode_1080p_force420.log
python tools/encode_time.py 1920 1080 95 try_420 2>&1 | tee $O/encode_1080p_try420.log
