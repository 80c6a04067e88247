#!/bin/bash
# The round's profile set of the FINAL code in one session (copied to profiles/rNN_* afterwards by
# tools/collect_profiles.sh): chain timings, per-kernel rocprofv3 statistics with the chain
# serialised, SQ counters incl. LDS bank conflicts, FETCH/WRITE traffic, the bench line, the
# rocprofv3 kernel statistics of the bench command, whole-encode timers, one iteration's timeline.
# Usage: tools/gpurun_head.sh --timeout 2400 -- 'bash tools/gpu_profiles.sh [tag]'
set -u
export TMPDIR=/tmp
TAG=${1:-prof}; O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cp .gpurun_head $O/head.txt 2>/dev/null || true
tools/ubench/bw 2>/dev/null | head -3 | tee $O/bw.log
{ echo "# head $(cat .gpurun_head 2>/dev/null)"; for i in 1 2 3; do python tools/run_compare.py 3840 2160 200; python tools/run_compare.py 1920 1080 400; done; } | tee $O/compare_chain.log
for sz in "3840 2160 20 4k" "1920 1080 40 1080p"; do set -- $sz
  d=$O/trace_$4
  ( cd /tmp && env GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$d -- python $R/tools/run_compare.py $1 $2 $3 ) > $d.log 2>&1
  f=$(grep -l "k_malta" $(find $d -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp $f $O/compare_$4_kernel_stats_single_stream.csv; rm -rf $d
done
bash tools/gpu_sq.sh $TAG/sq > $O/sq.log 2>&1; tail -20 $O/sq.log
bash tools/gpu_pmc.sh $TAG/pmc > $O/pmc.log 2>&1; tail -5 $O/pmc.log
{ echo "# head $(cat .gpurun_head 2>/dev/null)"; python tools/encode_time.py 1920 1080 95 6; python tools/encode_time.py 3840 2160 95 4; python tools/encode_time.py 1920 1080 95 force_420 4; python tools/encode_time.py 3840 2160 84 3; } 2>&1 | tee $O/encode_timers.log | cut -c1-200
python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-300
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/benchprof -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5 --batch-images 0 --batch-1mpix 0 ) > $O/benchprof.log 2>&1
# (bench.py runs tools/ubench in child processes: rocprofv3 writes one summary per process -- the
# bench's own is the one with the chain's kernels in it)
f=$(grep -l "k_malta" $(find $O/benchprof -name "*kernel_stats.csv") | head -1); [ -n "$f" ] && cp $f $O/bench_kernel_stats.csv; rm -rf $O/benchprof
bash tools/gpu_trace_full.sh $TAG/timeline > $O/timeline.log 2>&1
bash tools/gpu_trace_full.sh $TAG/timeline4k 3840 2160 > $O/timeline4k.log 2>&1
