#!/bin/bash
# Round 6, final validation: the GPU suite, the bench line (kept as profiles/r06_bench.json), the driver's smoke().
set -u
export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
( timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 ) | tee $O/gputests.log
python bench.py > $O/bench.json 2> $O/bench.err; tail -2 $O/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
