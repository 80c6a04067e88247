#!/bin/bash
# Entropy coder on its own stream beside Compare (single sync per scan), device-side ranking
# of the zeroing candidates: parity suite, encode timers, bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/s14; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2 | tee $O/bw.log
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python tools/encode_time.py 1920 1080 2>&1 | tail -2 | tee $O/encode_1080.log
python tools/encode_time.py 3840 2160 2>&1 | tail -2 | tee $O/encode_4k.log
python tools/encode_time.py 3840 2160 84 2>&1 | tail -2 | tee $O/encode_4k_q84.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
