#!/bin/bash
# Slow-box hunt: is this box one where the product's unrolled kernels are slow,
# and what distinguishes it (driver / firmware versions, A/B of the compact-code variants)?
set -u
export TMPDIR=/tmp
O=gpurun_out/s22; mkdir -p $O
TAG=$(date +%H%M%S)
{ tools/ubench/bw 2>/dev/null | head -2
  for v in "0 0" "1 1" "0 0" "1 1"; do set -- $v; echo "GZ_COMPACT_BLUR_V=$1 GZ_COMPACT_BLUR2D=$2"; GZ_COMPACT_BLUR_V=$1 GZ_COMPACT_BLUR2D=$2 python tools/run_compare.py 1920 1080 60; done
  echo "amdgpu module: $(cat /sys/module/amdgpu/version 2>/dev/null)"; uname -r
  rocm-smi --showfwinfo 2>/dev/null | grep -i "firmware version" | head -24
  rocm-smi --showvbios --showserial --showuniqueid 2>/dev/null | grep -i "vbios\|serial\|unique" | head -4
  cat /sys/class/kfd/kfd/topology/nodes/*/properties 2>/dev/null | grep -i "fw_version\|sdma_fw\|simd_count\|cu_count\|max_engine_clk_fcompute" | sort | uniq -c | head -10
} 2>&1 | tee $O/box_$TAG.log
