#!/bin/bash
# Round-end validation of HEAD: gpu suite, smoke, bench (all legs), rocprofv3 kernel-trace
# summaries of the bench command and of the Compare chain (3 streams and serialised).
set -u
export TMPDIR=/tmp
O=gpurun_out/s20; mkdir -p $O
R=$GRAFT_REPO_ROOT
tools/ubench/bw 2>/dev/null | head -2 | tee $O/bw.log
( timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.log
( timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cut -c1-200 $O/bench.json; tail -1 $O/bench.err
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace_bench -- python $R/bench.py --steps 3 --no-4k --no-cpu-baseline --batch-images 0 ) > $O/trace_bench.log 2>&1; grep -o '"ms_per_compare": [0-9.]*' $O/trace_bench.log | head -2
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace4k -- python $R/tools/run_compare.py 3840 2160 20 ) > $O/trace4k.log 2>&1; tail -1 $O/trace4k.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace1080 -- python $R/tools/run_compare.py 1920 1080 40 ) > $O/trace1080.log 2>&1; tail -1 $O/trace1080.log
( cd /tmp && GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace4k_ss -- python $R/tools/run_compare.py 3840 2160 20 ) > $O/trace4k_ss.log 2>&1; tail -1 $O/trace4k_ss.log
( cd /tmp && GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/trace1080_ss -- python $R/tools/run_compare.py 1920 1080 40 ) > $O/trace1080_ss.log 2>&1; tail -1 $O/trace1080_ss.log
find $O -name "*kernel_trace.csv" -delete; find $O -name "*agent_info.csv" -delete
