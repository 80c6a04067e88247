#!/bin/bash
# Whole-encode A/B of build-time variants of the device library (guetzli_amd/variants/<name>.so), two
# rounds, 4K and 1080p, with the host timers that matter for phase B.
# Usage: gpurun -- 'bash tools/gpu_encode_ab.sh TAG name1 name2 ...'
set -u
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
{
for rep in 1 2 3; do
  for v in "$@"; do
    cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd.so
    echo "== $v"
    python tools/encode_time.py 3840 2160 95 x 5 | python -c "
import sys,re,ast
t=sys.stdin.read()
print(t.splitlines()[0][:150])
m=re.search(r'timers: (\{.*\})', t)
d=ast.literal_eval(m.group(1))
print('  ', {k:d[k] for k in ('total','phase_b_host','pb_loop_fast_steps','pb_fast_apply','pb_fast_delta','pb_fast_count','compare','block_search') if k in d})"
    python tools/encode_time.py 1920 1080 95 x 5 | head -1 | cut -c1-110
  done
done
} 2>&1 | tee $O/encode_ab.log
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
