#!/bin/bash
# Compare chain as a captured graph vs plain launches; then the full gpu suite and the bench.
set -u
export TMPDIR=/tmp
O=gpurun_out/s13; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2 | tee $O/bw.log
for g in 0 1; do
  echo "== GZ_NO_GRAPH=$g" | tee -a $O/compare.log
  GZ_NO_GRAPH=$g python tools/run_compare.py 1920 1080 100 | tee -a $O/compare.log
  GZ_NO_GRAPH=$g python tools/run_compare.py 3840 2160 40 | tee -a $O/compare.log
  GZ_NO_GRAPH=$g python tools/encode_time.py 1920 1080 2>&1 | tail -2 | tee -a $O/compare.log
done
GZ_NO_GRAPH=0 python tools/encode_time.py 3840 2160 2>&1 | tail -2 | tee -a $O/compare.log
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
