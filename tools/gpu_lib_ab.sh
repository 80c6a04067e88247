#!/bin/bash
# A/B of several builds of the device library (tools/variants/*.so): chain time at 1080p and 4K.
# Usage: gpu_lib_ab.sh TAG "ENV=val ENV2=val" lib1.so lib2.so ...   (first argument after TAG: environment for all runs, may be "")
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; ENVS=$2; shift; shift
{
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib $ENVS"; env $ENVS GUETZLI_AMD_LIB=$PWD/$lib python tools/run_compare.py 1920 1080 100; env $ENVS GUETZLI_AMD_LIB=$PWD/$lib python tools/run_compare.py 3840 2160 40
done; done
} 2>&1 | tee -a $O/ab.log
