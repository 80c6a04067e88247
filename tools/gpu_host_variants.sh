#!/bin/bash
# A/B of builds of the HOST library (guetzli_amd/variants/<name>.so copied over
# libguetzli_amd_host.so in turn): whole-encode time and the phase-B timers.
# Usage: gpurun -- 'bash tools/gpu_host_variants.sh TAG name1 name2 ...'   (two repetitions)
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
cp guetzli_amd/libguetzli_amd_host.so /tmp/host_orig.so
{
for rep in 1 2; do
  for v in "$@"; do
    cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd_host.so
    echo "== $v"
    python tools/encode_time.py 1920 1080 95 x 4 | sed -e 's/iters.*//' -e 's/, .compare_begin.*//'
    python tools/encode_time.py 3840 2160 95 x 4 | sed -e 's/iters.*//' -e 's/, .compare_begin.*//'
  done
done
} 2>&1 | tee $O/host_variants.log
cp /tmp/host_orig.so guetzli_amd/libguetzli_amd_host.so
