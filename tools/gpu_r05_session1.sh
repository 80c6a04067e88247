#!/bin/bash
# Round 5, first GPU session: (1) the new photograph / quality goldens through the C ABI; (2) issue
# cost per instruction class and of the Malta line-sum pattern (tools/ubench/issue); (3) BASELINE
# config 5's single-GPU slice (8 x 4K, several in flight) under the knobs that change occupancy;
# (4) the counters of that batch -- which unit is busy while the GPU spends 0.22 s per image
# (VERDICT r4 item 1a): --pmc passes only (no tracing beside them), per-kernel sums written by
# tools/batch_pmc_summary.py.  Usage: tools/gpurun_head.sh --timeout 1500 -- 'bash tools/gpu_r05_session1.sh'
set -u
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null
T0=$(date +%s)
lap() { echo "== $1: $(( $(date +%s) - T0 )) s" | tee -a $O/laps.log; }

timeout 120 tools/ubench/issue > $O/issue.log 2>&1; lap issue
timeout 60 tools/ubench/bw > $O/bw.log 2>&1; lap bw
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "photo" > $O/pytest_photos.log 2>&1; lap pytest_photos
tail -3 $O/pytest_photos.log

# (3) batch variants: 8 x 4K, two timed repetitions each
for v in "4 _" "2 _" "6 _" "4 GZ_TILE_ROWS=16" "4 GZ_BLUR_PK=0" "4 GZ_SINGLE_STREAM=1"; do
  set -- $v
  envs=(); [ "$2" != "_" ] && envs=("$2")
  echo "## in flight $1, $2" >> $O/batch_variants.log
  env "${envs[@]}" timeout 200 python tools/batch_time.py 3840 2160 8 $1 2 >> $O/batch_variants.log 2>&1
done; lap batch_variants
cat $O/batch_variants.log

# chain alone, for the session's reference
for sz in "3840 2160" "1920 1080"; do timeout 100 python tools/run_compare.py $sz 200 >> $O/compare_chain.log 2>&1; done; lap chain
cat $O/compare_chain.log

# (4) counters of the batch (4 x 4K, 4 in flight; the profiler serialises the dispatches: the SUMS
# per image are what the real, co-running batch has to execute)
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
B="SQ_WAVES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
for p in A B; do
  eval ctrs=\$$p
  ( cd /tmp && timeout 400 rocprofv3 --pmc $ctrs --output-format csv -d $R/$O/batch_sq_$p -- python $R/tools/batch_time.py 3840 2160 4 4 0 ) > $O/batch_sq_$p.log 2>&1
  lap batch_sq_$p
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/batch_$ctr -- python $R/tools/batch_time.py 3840 2160 4 4 0 ) > $O/batch_$ctr.log 2>&1
  lap batch_$ctr
done
python tools/batch_pmc_summary.py $O/batch_sq_A $O/batch_sq_B $O/batch_FETCH_SIZE $O/batch_WRITE_SIZE > $O/config5_pmc_per_kernel.csv 2> $O/config5_pmc_summary.txt
cat $O/config5_pmc_summary.txt
# the same batch under the kernel trace (dispatches overlap here): summed kernel time per image
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/batch_trace -- python $R/tools/batch_time.py 3840 2160 8 4 1 ) > $O/batch_trace.log 2>&1
f=$(find $O/batch_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/config5_kernel_stats.csv
lap batch_trace
find $O -name "*counter_collection.csv" -delete
find $O -name "*kernel_trace.csv" -delete
find $O -name "*agent_info.csv" -delete
du -sh $O
