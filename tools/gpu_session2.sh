#!/bin/bash
set -u
O=gpurun_out/s2; mkdir -p $O
python tools/record_replay.py 444 258 95 $O/bees_q95.log 2>&1 | tail -1
python tools/record_replay.py 1920 1080 95 $O/hd_q95.log 2>&1 | tail -1
gzip -1 $O/hd_q95.log
ls -la $O
