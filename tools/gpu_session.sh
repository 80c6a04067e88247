#!/bin/bash
# One GPU session (run through gpurun): the -m gpu suite, a short bench, kernel statistics.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_session.sh [tag] [full]'
set -u
export TMPDIR=/tmp
TAG=${1:-s}
O=gpurun_out/$TAG; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 | tee $O/pytest.log
python bench.py --steps 10 --warmup 3 2>$O/bench.err | tee $O/bench.json
python tools/run_compare.py 3840 2160 30 | tee $O/compare_4k.log
python tools/run_compare.py 1920 1080 60 | tee $O/compare_1080.log
