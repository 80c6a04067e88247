#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s12; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for t in 16384 32768 65536 131072; do echo -n "thr $t: "; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 1920 1080 | tail -1 | grep -o "'total': [0-9.]*\|'pb_loop_ensure_sorted': [0-9.]*" | tr '\n' ' '; echo; done
for t in 32768 65536 131072 262144; do echo -n "4k thr $t: "; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 3840 2160 | tail -1 | grep -o "'total': [0-9.]*\|'pb_loop_ensure_sorted': [0-9.]*" | tr '\n' ' '; echo; done
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
