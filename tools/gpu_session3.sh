#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s3; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
( timeout 300 python tools/encode_time.py 1920 1080 > $O/encode_1080.log 2>&1; echo "rc=$?" >> $O/encode_1080.log ); cat $O/encode_1080.log
( timeout 600 python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; echo "rc=$?" >> $O/encode_4k.log ); cat $O/encode_4k.log
python tools/record_replay.py 1920 1080 95 $O/hd_q95.log 2>&1 | tail -1
gzip -1 $O/hd_q95.log
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o enc -- python tools/encode_time.py 1920 1080 > $O/prof_encode.log 2>&1 )
ls -la $O
