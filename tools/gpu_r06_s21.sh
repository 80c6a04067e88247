#!/bin/bash
# Round 6, session 21: patches behind the statistics (k_reconstruct_listed) -- A/B against the full reconstruction,
# the patch test, a timeline.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06x; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "patched or whole_encode_4k or self_checks" 2>&1 | tail -3 | tee $O/tests.log
{
for rep in 1 2 3; do
  for e in GZ_PATCH_RECON=1 GZ_PATCH_RECON=0; do
    echo "== $e"
    env $e python tools/encode_time.py 3840 2160 95 8 | head -1 | cut -c1-150
    env $e python tools/encode_time.py 1920 1080 95 10 | head -1 | cut -c1-150
    env $e python tools/encode_time.py 1024 1024 95 10 | head -1 | cut -c1-150
  done
done
for rep in 1 2; do
  for e in GZ_PATCH_RECON=1 GZ_PATCH_RECON=0; do
    echo "== $e"
    env $e python tools/batch_time.py 3840 2160 8 4 2
    env $e python tools/batch_time.py 1920 1080 16 4 2
    env $e python tools/batch_time.py 1024 1024 64 6 2
  done
done
} 2>&1 | tee $O/ab.log
bash tools/gpu_trace_full.sh r06x 3840 2160 > /dev/null
