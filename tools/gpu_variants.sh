#!/bin/bash
# A/B of BUILD-TIME variants of the device library (guetzli_amd/variants/<name>.so, made by
# `python -m guetzli_amd.build --variant NAME [DEFINE ...]`): each is copied over
# libguetzli_amd.so in turn and measured with the chain / block search / whole-encode tools.
# Usage: gpurun -- 'bash tools/gpu_variants.sh TAG name1 name2 ...'   (two repetitions)
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
{
for rep in 1 2; do
  for v in "$@"; do
    cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd.so
    echo "== $v"
    python tools/run_compare.py 1920 1080 100
    python tools/run_compare.py 3840 2160 40
    python tools/run_search.py 1920 1080 | tail -1
    python tools/run_search.py 3840 2160 | tail -1
    python tools/encode_time.py 1920 1080 95 x 4 | head -1 | cut -c1-120
    python tools/encode_time.py 3840 2160 95 x 4 | head -1 | cut -c1-120
  done
done
} 2>&1 | tee $O/variants.log
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
