#!/bin/bash
# Result mailbox (k_post + host polling) vs pinned copies: suite, timers, bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/s18; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2 | tee $O/bw.log
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
for m in 1 0; do
  echo "== GZ_NO_MAILBOX=$m" | tee -a $O/encode.log
  GZ_NO_MAILBOX=$m python tools/encode_time.py 1920 1080 2>&1 | tail -1 | cut -c1-400 | tee -a $O/encode.log
  GZ_NO_MAILBOX=$m python tools/encode_time.py 1920 1080 2>&1 | tail -2 | tee -a $O/encode.log
  GZ_NO_MAILBOX=$m python tools/encode_time.py 3840 2160 2>&1 | tail -2 | tee -a $O/encode.log
done
python tools/encode_time.py 3840 2160 84 2>&1 | tail -2 | tee -a $O/encode.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json | cut -c1-300; tail -2 $O/bench.err
