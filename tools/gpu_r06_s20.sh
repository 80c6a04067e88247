#!/bin/bash
# Round 6, session 20: the candidate's linear planes patched by the calls that change it (gz_config.patch_reconstruct)
# against the chain with its full reconstruction in front: GPU tests, whole encodes, batches.
set -u
export TMPDIR=/tmp
O=gpurun_out/r06v; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -5 | tee $O/tests.log
{
for rep in 1 2 3; do
  for e in GZ_PATCH_RECON=1 GZ_PATCH_RECON=0; do
    echo "== $e"
    env $e python tools/encode_time.py 3840 2160 95 6 | head -1 | cut -c1-150
    env $e python tools/encode_time.py 1920 1080 95 8 | head -1 | cut -c1-150
    env $e python tools/encode_time.py 1024 1024 95 8 | head -1 | cut -c1-150
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
