#!/bin/bash
set -u
export TMPDIR=/tmp
O=gpurun_out/s21; mkdir -p $O
python tools/encode_time.py 1920 1080 > /dev/null 2>&1
for t in 16384 32768 65536 131072 262144; do for rep in 1 2; do echo -n "thr $t: " | tee -a $O/thr.log; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 1920 1080 | tail -1 | grep -o "'select_frequency_masking': [0-9.]*\|'pb_loop_ensure_sorted': [0-9.]*" | tr '\n' ' ' | tee -a $O/thr.log; echo | tee -a $O/thr.log; done; done
for t in 32768 65536 131072 262144 524288; do echo -n "4k thr $t: " | tee -a $O/thr.log; GZ_ORDER_DEVICE_THRESHOLD=$t python tools/encode_time.py 3840 2160 | tail -1 | grep -o "'select_frequency_masking': [0-9.]*\|'pb_loop_ensure_sorted': [0-9.]*" | tr '\n' ' ' | tee -a $O/thr.log; echo | tee -a $O/thr.log; done
