#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s19; mkdir -p $O
python tools/time_create.py 2>&1 | tee $O/create.log
GZ_POOL_MB=0 python tools/time_create.py 2>&1 | tee -a $O/create.log
