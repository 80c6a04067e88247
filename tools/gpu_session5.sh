#!/bin/bash
# GPU session 5: gpu parity tests, bench, replay logs for host-side profiling off-box.
set -u
export TMPDIR=/tmp
O=gpurun_out/s5; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
python tools/record_replay.py 1920 1080 95 /tmp/r1080.log > $O/record_1080.log 2>&1; cat $O/record_1080.log
python tools/record_replay.py 3840 2160 95 /tmp/r4k.log > $O/record_4k.log 2>&1; cat $O/record_4k.log
ls -la /tmp/*.log
xz -T0 -3 -c /tmp/r1080.log > $O/r1080.log.xz; xz -T0 -3 -c /tmp/r4k.log > $O/r4k.log.xz; ls -la $O
python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; cat $O/encode_4k.log
nproc
