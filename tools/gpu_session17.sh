#!/bin/bash
# Experiments: 16-row blur tiles at 1080p/4K, histogram grid, device-order threshold.
set -u
export TMPDIR=/tmp
O=gpurun_out/s17; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -2 | tee $O/bw.log
for t in 32 16; do
  echo "== GZ_TILE_ROWS=$t" | tee -a $O/tiles.log
  GZ_TILE_ROWS=$t python tools/run_compare.py 1920 1080 100 | tee -a $O/tiles.log
  GZ_TILE_ROWS=$t python tools/run_compare.py 3840 2160 40 | tee -a $O/tiles.log
  GZ_TILE_ROWS=$t python tools/run_compare.py 1280 720 100 | tee -a $O/tiles.log
done
for g in 512 1024 2048 4096; do echo -n "hist grid $g: " | tee -a $O/hist.log; GZ_HIST_GRID=$g python tools/encode_time.py 1920 1080 | tail -1 | grep -o "'total': [0-9.]*\|'pb_loop_fast_steps': [0-9.]*" | tr '\n' ' ' | tee -a $O/hist.log; echo | tee -a $O/hist.log; done
for t in 8192 16384 32768 65536 131072; do echo -n "thr $t: " | tee -a $O/thr.log; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 1920 1080 | tail -1 | grep -o "'total': [0-9.]*\|'pb_loop_ensure_sorted': [0-9.]*" | tr '\n' ' ' | tee -a $O/thr.log; echo | tee -a $O/thr.log; done
for t in 16384 32768 65536 131072 262144; do echo -n "4k thr $t: " | tee -a $O/thr.log; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 3840 2160 | tail -1 | grep -o "'total': [0-9.]*\|'pb_loop_ensure_sorted': [0-9.]*" | tr '\n' ' ' | tee -a $O/thr.log; echo | tee -a $O/thr.log; done
GZ_TILE_ROWS=16 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or stages or compare or diffmap or encode_matches" 2>&1 | tail -2
