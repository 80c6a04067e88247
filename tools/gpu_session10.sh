#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s10; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2
( timeout 1200 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -5 $O/pytest.log
python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; tail -1 $O/encode_4k.log
python tools/encode_time.py 3840 2160 84 > $O/encode_4k_q84.log 2>&1; tail -1 $O/encode_4k_q84.log
python tools/encode_time.py 1920 1080 > $O/encode_1080.log 2>&1; tail -1 $O/encode_1080.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
bash tools/gpu_pmc.sh > $O/pmc.log 2>&1; tail -3 $O/pmc.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace4k -- python $GRAFT_REPO_ROOT/tools/run_compare.py 3840 2160 20 ) > $O/trace4k.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1080 -- python $GRAFT_REPO_ROOT/tools/run_compare.py 1920 1080 40 ) > $O/trace1080.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace_bench -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --batch-images 0 ) > $O/trace_bench.log 2>&1
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
