#!/bin/bash
# Copies the summaries of a tools/gpu_profiles.sh session (gpurun_out/<tag>) into profiles/<round>_*,
# every file stamped with the commit the session ran (head.txt).  Usage: collect_profiles.sh TAG ROUND   (e.g. prof r04)
set -eu
cd "$(dirname "$0")/.."
T=gpurun_out/${1:-prof}; P=profiles; RN=${2:-r04}
H=$(cat $T/head.txt 2>/dev/null || echo unknown)
stamp() { { echo "# head $H ($3)"; cat "$1"; } > "$2"; }
stamp $T/compare_4k_kernel_stats_single_stream.csv $P/${RN}_compare_4k_kernel_stats_single_stream.csv "rocprofv3 --kernel-trace --stats, GZ_SINGLE_STREAM=1, tools/run_compare.py 3840 2160 20"
stamp $T/compare_1080p_kernel_stats_single_stream.csv $P/${RN}_compare_1080p_kernel_stats_single_stream.csv "rocprofv3 --kernel-trace --stats, GZ_SINGLE_STREAM=1, tools/run_compare.py 1920 1080 40"
stamp $T/sq/chain_sq_4k.csv $P/${RN}_compare_4k_sq_counters.csv "rocprofv3 --pmc SQ_* (three passes), GZ_SINGLE_STREAM=1, tools/gpu_sq.sh"
stamp $T/sq/chain_sq_1080.csv $P/${RN}_compare_1080p_sq_counters.csv "rocprofv3 --pmc SQ_* (three passes), GZ_SINGLE_STREAM=1, tools/gpu_sq.sh"
stamp $T/pmc/compare_4k_pmc.csv $P/${RN}_compare_4k_pmc.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/gpu_pmc.sh"
stamp $T/pmc/compare_1080p_pmc.csv $P/${RN}_compare_1080p_pmc.csv "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/gpu_pmc.sh"
stamp $T/pmc/block_search_pmc.csv $P/${RN}_block_search_pmc.csv "rocprofv3 --pmc SQ_* of k_block_search<0>, tools/run_search.py 1920 1080"
EV=$(grep -o "[0-9]* CompareBlock evaluations" $T/pmc/search_1080p.log | head -1 | cut -d" " -f1)
[ -n "$EV" ] && sed -i "1a # evaluations per launch $EV" $P/${RN}_block_search_pmc.csv
python3 tools/pmc_traffic_json.py $T/pmc/compare_4k_pmc.csv $T/pmc/compare_1080p_pmc.csv $T/pmc/bw_pmc.csv | python3 -c "import json,sys; d=json.load(sys.stdin); d['head']='$H'; print(json.dumps(d, indent=1))" > $P/${RN}_compare_pmc_traffic.json
stamp $T/bench_kernel_stats.csv $P/${RN}_bench_kernel_stats.csv "rocprofv3 --kernel-trace --stats -- python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-config5 --batch-images 0"
python3 -c "import json; d=json.load(open('$T/bench.json')); d['head']='$H'; print(json.dumps(d))" > $P/${RN}_bench.json
cp $T/compare_chain.log $P/${RN}_compare_chain.log
cp $T/encode_timers.log $P/${RN}_encode_timers.log
stamp $T/timeline/timeline_full.txt $P/${RN}_encode_1080p_iteration_timeline_with_host_calls.txt "rocprofv3 --kernel-trace --memory-copy-trace --hip-runtime-trace, one phase-B iteration of a 1080p encode, tools/gpu_trace_full.sh"
stamp $T/timeline4k/timeline_full.txt $P/${RN}_encode_4k_iteration_timeline_with_host_calls.txt "the same of a 3840x2160 encode"
{ echo "# head $H"; bash tools/kernel_sizes.sh 2>/dev/null; } > $P/${RN}_kernel_code_sizes.csv
# round 5-6 extras (tools/gpu_profiles_r06.sh)
[ -f $T/compare_kernels.json ] && python3 -c "import json; d=json.load(open('$T/compare_kernels.json')); d['head']='$H'; print(json.dumps(d, indent=1))" > $P/${RN}_compare_kernels.json
[ -f $T/config5_pmc_summary.txt ] && { stamp $T/config5_pmc_summary.txt $P/${RN}_config5_pmc_summary.txt "rocprofv3 --pmc passes over tools/batch_time.py 3840 2160 4 4 0, summed by tools/batch_pmc_summary.py"; stamp $T/config5_pmc_per_kernel.csv $P/${RN}_config5_pmc_per_kernel.csv "the same, per kernel"; }
[ -f $T/issue.log ] && stamp $T/issue.log $P/${RN}_issue_cost.log "tools/ubench/issue on the MI355X"
[ -f $T/chain_in_process.log ] && stamp $T/chain_in_process.log $P/${RN}_chain_in_process.log "tools/chain_in_process.py 3840 2160"
for s in 4k 1080p; do [ -s $T/reference_cpu_$s.json ] && cp $T/reference_cpu_$s.json $P/${RN}_reference_cpu_$s.json; done
ls -la $P/${RN}_*
