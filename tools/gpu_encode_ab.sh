#!/bin/bash
# A/B of builds of the device library on whole encodes (median of N, hashes shown).
# Usage: gpu_encode_ab.sh TAG lib1.so lib2.so ...
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
LIBS=("$@")
{
for rep in 1 2; do for lib in "${LIBS[@]}"; do
  echo "== $lib"
  GUETZLI_AMD_LIB=$PWD/$lib python tools/encode_time.py 1920 1080 95 7 | head -1 | cut -c1-135
  GUETZLI_AMD_LIB=$PWD/$lib python tools/encode_time.py 3840 2160 95 3 | head -1 | cut -c1-135
done; done
GUETZLI_AMD_LIB=$PWD/${LIBS[-1]} python tools/encode_time.py 1920 1080 95 force_420 3 | head -1 | cut -c1-135
} 2>&1 | tee $O/ab.log
