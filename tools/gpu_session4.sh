#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s4; mkdir -p $O
( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -3 $O/pytest.log
( timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?" >> $O/bench.err ); cat $O/bench.json; tail -2 $O/bench.err
g++ -O2 -std=c++17 -pthread tools/bench_lazy_sort.cc -o /tmp/bls && for t in 1 4 8 16 32; do GZ_HOST_THREADS=$t /tmp/bls | tail -1; done > $O/lazy_sort_threads.log 2>&1; cat $O/lazy_sort_threads.log
for t in 1 4 8 16 32; do echo "threads $t"; GZ_HOST_THREADS=$t python tools/encode_time.py 1920 1080; done > $O/encode_threads.log 2>&1; grep -E "threads|timers" $O/encode_threads.log
python tools/encode_time.py 3840 2160 > $O/encode_4k.log 2>&1; cat $O/encode_4k.log
python tools/encode_time.py 3840 2160 84 > $O/encode_4k_q84.log 2>&1; cat $O/encode_4k_q84.log
nproc; lscpu | grep -E "Model name|Socket|Thread" 
# kernel trace + PMC passes of the Compare chain at 4K (working set 1.4 GB >> 256 MB infinity cache)
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace4k -- python $GRAFT_REPO_ROOT/tools/run_compare.py 3840 2160 20 ) > $O/trace4k.log 2>&1; tail -1 $O/trace4k.log
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_fetch -- python $GRAFT_REPO_ROOT/tools/run_compare.py 3840 2160 3 ) > $O/pmc_fetch.log 2>&1; tail -1 $O/pmc_fetch.log
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc_write -- python $GRAFT_REPO_ROOT/tools/run_compare.py 3840 2160 3 ) > $O/pmc_write.log 2>&1; tail -1 $O/pmc_write.log
python tools/pmc_summary.py $O/pmc_fetch $O/pmc_write > $O/pmc_summary.csv; head -50 $O/pmc_summary.csv
find $O -name "*.csv" | head -20
