#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s9; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; tail -1 $O/encode_4k.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
for w in 2 8; do timeout 300 python bench.py --no-cpu-baseline --steps 1 --warmup 0 --batch-images 16 --batch-workers $w 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('workers', $w, d['batch_one_gpu'])"; done
