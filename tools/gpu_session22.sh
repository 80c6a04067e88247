#!/bin/bash
# Slow-box hunt: is this box one where the tiled kernels are slow, and does memory mapped through
# the VMM API behave differently there?
set -u
export TMPDIR=/tmp
O=gpurun_out/s22; mkdir -p $O
TAG=$(date +%H%M%S)
{ tools/ubench/bw 2>/dev/null | head -2; python tools/run_compare.py 1920 1080 60; python tools/run_compare.py 3840 2160 30; tools/ubench/tile; rocm-smi --showmemorypartition --showcomputepartition 2>&1 | grep -i partition | head -4; rocm-smi --showclocks 2>&1 | grep -i 'sclk\|mclk\|fclk' | head -4; rocminfo 2>/dev/null | grep -i 'xnack\|Compute Unit\|Max Clock' | head -6; } 2>&1 | tee $O/box_$TAG.log
