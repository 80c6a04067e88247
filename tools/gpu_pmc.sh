#!/bin/bash
# PMC passes (separate runs, --pmc only, no tracing): FETCH_SIZE and WRITE_SIZE of the Compare
# chain at 4K and 1080p with the streaming-copy micro-benchmark as calibration of the counters;
# SQ issue counters of the block search kernel (VALU utilisation).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-pmc}; mkdir -p $O
R=$GRAFT_REPO_ROOT
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/c4k_$ctr -- python $R/tools/run_compare.py 3840 2160 3 ) > $O/c4k_$ctr.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/c1080_$ctr -- python $R/tools/run_compare.py 1920 1080 3 ) > $O/c1080_$ctr.log 2>&1
  ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --output-format csv -d $R/$O/bw_$ctr -- $R/tools/ubench/bw ) > $O/bw_$ctr.log 2>&1
done
python tools/pmc_summary.py $O/c4k_FETCH_SIZE $O/c4k_WRITE_SIZE > $O/compare_4k_pmc.csv
python tools/pmc_summary.py $O/c1080_FETCH_SIZE $O/c1080_WRITE_SIZE > $O/compare_1080p_pmc.csv
python tools/pmc_summary.py $O/bw_FETCH_SIZE $O/bw_WRITE_SIZE > $O/bw_pmc.csv
cat $O/bw_pmc.csv
python tools/pmc_traffic_json.py $O/compare_4k_pmc.csv $O/compare_1080p_pmc.csv $O/bw_pmc.csv > $O/traffic.json; head -24 $O/traffic.json
python tools/run_search.py 1920 1080 | tee $O/search_1080p.log
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/search_sq -- python $R/tools/run_search.py 1920 1080 ) > $O/search_sq.log 2>&1
python tools/pmc_summary.py $O/search_sq | grep -i "block_search\|^kernel" | tee $O/block_search_pmc.csv
find $O -name "*counter_collection.csv" -delete   # keep the summaries only (size)
