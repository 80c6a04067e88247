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
