#!/bin/bash
# A/B of a Compare-chain knob given as VAR=a,b (default GZ_MASK_SPLIT=0,1): chain time at 4K and 1080p.
set -u
export TMPDIR=/tmp
KV=${1:-GZ_MASK_SPLIT=0,1}; VAR=${KV%%=*}; VALS=${KV#*=}
O=gpurun_out/${2:-chainab}; mkdir -p $O
{
for rep in 1 2 3; do for v in ${VALS//,/ }; do
  echo "== $VAR=$v"; env $VAR=$v python tools/run_compare.py 1920 1080 100; env $VAR=$v python tools/run_compare.py 3840 2160 40
done; done
for v in ${VALS//,/ }; do echo "== encode $VAR=$v"; env $VAR=$v python tools/encode_time.py 1920 1080 95 6 | head -1 | cut -c1-120; done
} 2>&1 | tee $O/ab.log
