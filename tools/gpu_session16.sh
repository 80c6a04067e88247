#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s16; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2 | tee $O/bw.log
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
python tools/encode_time.py 1920 1080 2>&1 | tail -2 | tee $O/encode_1080.log
python tools/encode_time.py 1920 1080 2>&1 | tail -2 | tee -a $O/encode_1080.log
python tools/encode_time.py 3840 2160 2>&1 | tail -2 | tee $O/encode_4k.log
python tools/encode_time.py 3840 2160 84 2>&1 | tail -2 | tee $O/encode_4k_q84.log
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1080 -- python $GRAFT_REPO_ROOT/tools/encode_time.py 1920 1080 ) > $O/trace1080.log 2>&1; tail -1 $O/trace1080.log
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; cut -d, -f1-4 $f | sed 's/gz:://g' | cut -c1-150 | head -40; done
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json | cut -c1-400; tail -2 $O/bench.err
