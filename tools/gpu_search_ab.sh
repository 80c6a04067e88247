#!/bin/bash
# A/B of build-time variants of the block-search kernel (guetzli_amd/variants/<name>.so): parity of the
# tree's library first (the search's GPU tests), then per variant the search alone and whole encodes.
# Usage: gpurun -- 'bash tools/gpu_search_ab.sh TAG name1 name2 ...'
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search or zeroing or compare_block or seam or 420 or golden" 2>&1 | tail -4 | tee $O/pytest.log
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
{
for rep in 1 2; do
  for v in "$@"; do
    cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd.so
    echo "== $v"
    python tools/run_search.py 1920 1080 | tail -1
    python tools/run_search.py 3840 2160 | tail -1
    python tools/encode_time.py 3840 2160 95 x 4 | tail -1 | tr ',' '\n' | grep -i "'block_search'\|'total'" | tr '\n' ' '; echo
  done
done
} 2>&1 | tee $O/variants.log
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
# counters of the tree's library (VALU instructions per evaluation, busy / wait cycles)
R=$(pwd); export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS --output-format csv -d $R/$O/search_sq -- python $R/tools/run_search.py 1920 1080 ) > $O/search_sq.log 2>&1
python tools/pmc_summary.py $O/search_sq | grep -i "block_search\|^kernel" | tee $O/block_search_pmc.csv
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/search_trace -- python $R/tools/run_search.py 3840 2160 ) > $O/search_trace.log 2>&1
f=$(find $O/search_trace -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { grep -i "search\|rank\|csr\|Name" $f | cut -d, -f1-4 | sed 's/gz:://g' | cut -c1-110 | tee $O/search_kernel_stats.txt; }
rm -rf $O/search_sq $O/search_trace
