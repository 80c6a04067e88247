#!/bin/bash
# A/B of phase B's order construction enqueued ahead (gz_order_build_auto_begin/_end, GZ_ORDER_AHEAD).
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-ahead}; mkdir -p $O
{
python -m pytest tests/test_gpu_parity.py -x -q -k "global_order or device_partition" 2>&1 | tail -3
for rep in 1 2 3; do for a in 0 1; do
  echo "== GZ_ORDER_AHEAD=$a"; GZ_ORDER_AHEAD=$a python tools/encode_time.py 1920 1080 95 4
done; done
for a in 0 1; do echo "== 4K GZ_ORDER_AHEAD=$a"; GZ_ORDER_AHEAD=$a python tools/encode_time.py 3840 2160 95 2; done
for a in 0 1; do echo "== 1080p 420 GZ_ORDER_AHEAD=$a"; GZ_ORDER_AHEAD=$a python tools/encode_time.py 1920 1080 95 force_420 3; done
} 2>&1 | tee $O/ahead_ab.log
