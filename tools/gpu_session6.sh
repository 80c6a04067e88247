#!/bin/bash
# GPU session 6: device global order (phase B) -- parity, encode timings, bench
set -u
export TMPDIR=/tmp
O=gpurun_out/s6; mkdir -p $O
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
python tools/encode_time.py 1920 1080 > $O/encode_1080.log 2>&1; cat $O/encode_1080.log
python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; cat $O/encode_4k.log
for t in 16384 32768 131072 262144; do echo "thr $t"; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 3840 2160 | tail -1; done > $O/encode_4k_thr.log 2>&1; cat $O/encode_4k_thr.log
for t in 4 8 32; do echo "threads $t"; GZ_HOST_THREADS=$t python tools/encode_time.py 3840 2160 | tail -1; done > $O/encode_4k_threads.log 2>&1; cat $O/encode_4k_threads.log
python tools/encode_time.py 3840 2160 84 > $O/encode_4k_q84.log 2>&1; cat $O/encode_4k_q84.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
