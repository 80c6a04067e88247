#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s8; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
GZ_VERIFY_ENTROPY=1 python tools/encode_time.py 1920 1080 2>&1 | tail -3 | cut -c1-400
python tools/encode_time.py 1920 1080 > $O/encode_1080.log 2>&1; cat $O/encode_1080.log
python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; cat $O/encode_4k.log
python tools/encode_time.py 3840 2160 84 > $O/encode_4k_q84.log 2>&1; cat $O/encode_4k_q84.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
python tools/record_replay.py 1920 1080 95 /tmp/r1080.log > $O/record_1080.log 2>&1; cat $O/record_1080.log
python tools/record_replay.py 3840 2160 95 /tmp/r4k.log > $O/record_4k.log 2>&1; cat $O/record_4k.log
xz -T0 -3 -c /tmp/r1080.log > $O/r1080.log.xz; xz -T0 -3 -c /tmp/r4k.log > $O/r4k.log.xz; ls -la $O | head -20
