#!/bin/bash
# Slow-box hunt: is this box one where the product's tiled kernels are slow, what do the
# micro-benchmarks say there, and what do the SQ counters of the Compare chain look like?
set -u
export TMPDIR=/tmp
O=gpurun_out/s22; mkdir -p $O
R=$GRAFT_REPO_ROOT
TAG=$(date +%H%M%S)
{ tools/ubench/bw 2>/dev/null | head -2; for v in "0 0" "1 0" "0 1" "1 1" "0 0" "1 1"; do set -- $v; echo "GZ_COMPACT_BLUR_V=$1 GZ_COMPACT_BLUR2D=$2"; GZ_COMPACT_BLUR_V=$1 GZ_COMPACT_BLUR2D=$2 python tools/run_compare.py 1920 1080 60; GZ_COMPACT_BLUR_V=$1 GZ_COMPACT_BLUR2D=$2 python tools/run_compare.py 3840 2160 30; done; } 2>&1 | tee $O/box_$TAG.log
