#!/bin/bash
# Round 6, session 1: GPU parity suite on the round's first changes, the CU-mask map of the part
# (tools/ubench/cumask), the chain with / without the distance map's store, batch mode on
# CU-partitioned stream sets (GZ_CU_PARTITION).   gpurun -- 'bash tools/gpu_r06_s1.sh'
set -u
export TMPDIR=/tmp
O=gpurun_out/r06a; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
( timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -5 ) | tee $O/gputests.log
{
  tools/ubench/cumask
  tools/ubench/cumask map 0 64
  tools/ubench/cumask map 64 128
  tools/ubench/cumask map 0 128
  tools/ubench/cumask map 0 192
  tools/ubench/cumask pair 0 128 128 256
  tools/ubench/cumask pair 0 64 64 128
} 2>&1 | tee $O/cumask.log
{
for rep in 1 2; do
  for cfg in "GZ_NONE=1" "GZ_STORE_DISTMAP=1" "GZ_CU_MAIN=0:192 GZ_CU_SIDE=192:256" "GZ_CU_MAIN=0:256 GZ_CU_SIDE=160:256" "GZ_CU_SIDE=128:256"; do
    echo "== $cfg"
    env $cfg python tools/run_compare.py 3840 2160 100
    env $cfg python tools/run_compare.py 1920 1080 200
  done
done
} 2>&1 | tee $O/chain.log
{
for rep in 1 2; do
  for cfg in "GZ_NONE=1" "GZ_CU_PARTITION=4" "GZ_CU_PARTITION=2" "GZ_STORE_DISTMAP=1"; do
    echo "== $cfg"
    env $cfg python tools/batch_time.py 3840 2160 8 4 2
    env $cfg python tools/batch_time.py 1920 1080 16 4 2
  done
done
echo "== GZ_CU_PARTITION=8, 8 in flight"
GZ_CU_PARTITION=8 python tools/batch_time.py 3840 2160 8 8 2
GZ_CU_PARTITION=8 python tools/batch_time.py 1920 1080 16 8 2
echo "== GZ_CU_PARTITION=2, 2 in flight"
GZ_CU_PARTITION=2 python tools/batch_time.py 3840 2160 8 2 2
} 2>&1 | tee $O/batch.log
