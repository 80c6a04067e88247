#!/bin/bash
# One GPU-box session: parity tests, bench line, rocprof kernel stats, stage timings.
set -u
export TMPDIR=/tmp
O=gpurun_out/s1; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q -k "not 1080p_bit_identical" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log )
tail -5 $O/pytest.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err )
cat $O/bench.json
( timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof -o bench -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $O/prof_bench.json 2> $O/prof.err; echo "prof rc=$?" >> $O/prof.err )
find $O/prof -name '*stats*' | head
( timeout 300 python tools/time_stages.py 1920 1080 > $O/stages_1080.log 2>&1 ); cat $O/stages_1080.log
( timeout 300 python tools/time_stages.py 3840 2160 > $O/stages_4k.log 2>&1 ); cat $O/stages_4k.log
( timeout 900 python tools/encode_time.py 1920 1080 > $O/encode_1080.log 2>&1; echo "rc=$?" >> $O/encode_1080.log ); cat $O/encode_1080.log
