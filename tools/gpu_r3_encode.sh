#!/bin/bash
# Round-3 encode session: the -m gpu suite, the chain with / without Malta's LDS-DMA staging,
# per-kernel statistics of the default chain, whole encodes with / without the device-decided
# quick-select descent (GZ_ORDER_DESCEND), and the default bench.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r3_encode.sh [tag] [skip]'
set -u
export TMPDIR=/tmp
TAG=${1:-e1}; O=gpurun_out/$TAG; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -3 | tee $O/bw.log
if [ "${2:-}" != "skip" ]; then
  ( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
fi
{
for rep in 1 2; do
  for cfg in "GZ_MALTA_DMA=0" "GZ_MALTA_DMA=1"; do
    echo "== $cfg"; env $cfg python tools/run_compare.py 1920 1080 100; env $cfg python tools/run_compare.py 3840 2160 40
  done
done
} 2>&1 | tee $O/chain_ab.log
for cfg in "GZ_MALTA_DMA=0" "GZ_MALTA_DMA=1"; do
  t=$(echo $cfg | tr -d ' =A-Z_')
  for sz in "3840 2160 20" "1920 1080 40"; do
    d=$O/trace_${t}_$(echo $sz | cut -d' ' -f1)
    ( cd /tmp && env $cfg GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/tools/run_compare.py $sz ) > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; }
  done
done
{
for rep in 1 2; do
  for cfg in "GZ_ORDER_DESCEND=0" "GZ_ORDER_DESCEND=1"; do
    echo "== $cfg"; env $cfg python tools/encode_time.py 1920 1080 95 6; env $cfg python tools/encode_time.py 3840 2160 95 3
  done
done
echo "== 420"; python tools/encode_time.py 1920 1080 95 force_420 4
} 2>&1 | tee $O/encode_ab.log | cut -c1-400
python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-400
