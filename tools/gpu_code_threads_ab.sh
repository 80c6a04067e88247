#!/bin/bash
# A/B of the number of code-refresh helper threads (GZ_CODE_THREADS, guetzli_amd/host/code_refresh.h):
# whole encodes at 4K and 1080p (median of the runs after the first), the host timers that change, and
# the config-5 slice (8 x 4K, 4 in flight).  Usage: gpurun -- 'bash tools/gpu_code_threads_ab.sh TAG 0 1 2 3'
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
{
for rep in 1 2; do
  for t in "$@"; do
    echo "== GZ_CODE_THREADS=$t"
    for sz in "3840 2160" "1920 1080"; do
      GZ_CODE_THREADS=$t python tools/encode_time.py $sz 95 x 7 | python -c "
import sys,re,ast
t=sys.stdin.read()
print(t.splitlines()[0][:62], re.search(r'in ([0-9.]+ s)', t).group(1), re.search(r\"'phase B steps taken ahead and undone': \d+\", t).group(0))
d=ast.literal_eval(re.search(r'timers: (\{.*\})', t).group(1))
print('  ', {k:d[k] for k in ('total','phase_b_host','pb_loop','pb_loop_codes','pb_loop_ensure_sorted','compare') if k in d})"
    done
    GZ_CODE_THREADS=$t python tools/batch_time.py 3840 2160 8 4 2 | tail -1
  done
done
} 2>&1 | tee $O/code_threads.log
