#!/bin/bash
# Round 6's profile set of the FINAL code in one session: everything tools/gpu_profiles.sh takes, plus
# the batch-mode counters (config 5's slice: which unit is busy), the per-kernel roofline JSON that
# bench.py puts on its line, the issue-cost table, and -- in the background on one pinned host core for
# the whole session -- the unmodified reference on the headline image (9 minutes; 1080p behind it).
# Usage: tools/gpurun_head.sh --timeout 2400 -- 'bash tools/gpu_profiles_r06.sh'; then tools/collect_profiles.sh r06p r05
set -u
export TMPDIR=/tmp
TAG=r06p; O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
# (SKIP_REF=1: a repeat of the set after a kernel change -- the reference's timing does not depend on it)
if [ "${SKIP_REF:-0}" = 1 ]; then ( true ) & else
( taskset -c 2 python tools/ref_cpu_time.py 3840 2160 > $O/reference_cpu_4k.json 2> $O/reference_cpu_4k.err;
  taskset -c 2 python tools/ref_cpu_time.py 1920 1080 > $O/reference_cpu_1080p.json 2> $O/reference_cpu_1080p.err ) &
fi
REFPID=$!
timeout 200 tools/ubench/issue > $O/issue.log 2>&1
# round 6: GPU suite on the final sources, the in-flight sweep for small images, the host's CPU quota
( timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -4 ) | tee $O/gputests.log
{ cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; python -c "import os; print(len(os.sched_getaffinity(0)))"; } > $O/host_cpus.log 2>&1
# one stream per image against three, batch mode (the default takes one when several contexts are alive)
{ for cfg in "GZ_NONE=1" "GZ_SINGLE_STREAM=0"; do echo "== $cfg"; for sz in "512 512 128" "1024 1024 64" "1920 1080 16" "3840 2160 8"; do env $cfg python tools/batch_time.py $sz 4 2; done; done
  echo "== GZ_NONE=1, 6 in flight"; python tools/batch_time.py 1024 1024 64 6 2; } 2>&1 | tee $O/batch_streams_default.log
bash tools/gpu_profiles.sh $TAG
# batch-mode counters (4 x 4K, 4 in flight) and the batch's own kernel statistics
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
B="SQ_WAVES SQ_WAIT_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM"
# (round 6: with one stream per image -- the default in batch mode -- the rocprofv3 --pmc passes of the batch did not
# finish in the session of the first attempt; every pass is tried in the default mode with a short limit and, if that
# fails, with the three-stream chain it was measured on until round 5 -- the summary says which)
BMODE=default
for p in A B; do
  eval ctrs=\$$p
  ( cd /tmp && timeout 200 rocprofv3 --pmc $ctrs --output-format csv -d $R/$O/batch_sq_$p -- python $R/tools/batch_time.py 3840 2160 4 4 0 ) > $O/batch_sq_$p.log 2>&1 || {
    BMODE=three_streams; rm -rf $O/batch_sq_$p
    ( cd /tmp && GZ_SINGLE_STREAM=0 timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/$O/batch_sq_$p -- python $R/tools/batch_time.py 3840 2160 4 4 0 ) > $O/batch_sq_$p.log 2>&1; }
done
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 200 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/batch_$ctr -- python $R/tools/batch_time.py 3840 2160 4 4 0 ) > $O/batch_$ctr.log 2>&1 || {
    BMODE=three_streams; rm -rf $O/batch_$ctr
    ( cd /tmp && GZ_SINGLE_STREAM=0 timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/batch_$ctr -- python $R/tools/batch_time.py 3840 2160 4 4 0 ) > $O/batch_$ctr.log 2>&1; }
done
echo "batch PMC passes: $BMODE" | tee $O/batch_pmc_mode.txt
python tools/batch_time.py 3840 2160 8 4 3 | tee $O/batch_time.log
sec=$(python3 -c "
import re,sys
r=[float(x) for x in re.findall(r'([0-9.]+)', open('$O/batch_time.log').read().split(':')[-1])]
print(round(3840*2160/1e6/(sum(r)/len(r)), 4))")
python tools/batch_pmc_summary.py $O/batch_sq_A $O/batch_sq_B $O/batch_FETCH_SIZE $O/batch_WRITE_SIZE --images 4 --seconds-per-image $sec > $O/config5_pmc_per_kernel.csv 2> $O/config5_pmc_summary.txt
cat $O/config5_pmc_summary.txt
find $O -name "*counter_collection.csv" -delete
python tools/kernel_roofline.py --stats4k $O/compare_4k_kernel_stats_single_stream.csv --stats1080 $O/compare_1080p_kernel_stats_single_stream.csv \
  --pmc4k $O/pmc/compare_4k_pmc.csv --pmc1080 $O/pmc/compare_1080p_pmc.csv --sq4k $O/sq/chain_sq_4k.csv --sq1080 $O/sq/chain_sq_1080.csv > $O/compare_kernels.json
python tools/chain_in_process.py > $O/chain_in_process.log 2>&1
echo "waiting for the reference on the host CPU"; wait $REFPID
[ -f $O/reference_cpu_4k.json ] && head -5 $O/reference_cpu_4k.json
du -sh $O
