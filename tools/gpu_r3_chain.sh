#!/bin/bash
# Round-3 chain session: the -m gpu suite, then the Compare chain under the round's code-path knobs
# (GZ_BLUR_OPT bits: 1 = epilogue by quads, 2 = conflict-free row-pass mappings; GZ_IDCT_DOT2),
# per-kernel rocprofv3 statistics with the chain serialised, and the default bench.
# Usage: gpurun --timeout 1500 -- 'bash tools/gpu_r3_chain.sh [tag] [skip-pytest]'
set -u
export TMPDIR=/tmp
TAG=${1:-c1}; O=gpurun_out/$TAG; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
tools/ubench/bw 2>/dev/null | head -3 | tee $O/bw.log
if [ "${2:-}" != "skip" ]; then
  ( timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
fi
{
for rep in 1 2; do
  for cfg in "GZ_BLUR_OPT=0 GZ_IDCT_DOT2=0" "GZ_BLUR_OPT=1 GZ_IDCT_DOT2=0" "GZ_BLUR_OPT=2 GZ_IDCT_DOT2=0" "GZ_BLUR_OPT=3 GZ_IDCT_DOT2=0" "GZ_BLUR_OPT=0 GZ_IDCT_DOT2=1" "GZ_BLUR_OPT=3 GZ_IDCT_DOT2=1"; do
    echo "== $cfg"; env $cfg python tools/run_compare.py 1920 1080 100; env $cfg python tools/run_compare.py 3840 2160 40
  done
done
} 2>&1 | tee $O/chain_ab.log
for cfg in "GZ_BLUR_OPT=0 GZ_IDCT_DOT2=0" "GZ_BLUR_OPT=3 GZ_IDCT_DOT2=1"; do
  t=$(echo $cfg | tr -d ' =A-Z_')
  for sz in "3840 2160 20" "1920 1080 40"; do
    d=$O/trace_${t}_$(echo $sz | cut -d' ' -f1)
    ( cd /tmp && env $cfg GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/tools/run_compare.py $sz ) > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1)
    [ -n "$f" ] && { echo "== $cfg $sz"; cut -d, -f1-4 $f | sed 's/gz:://g' | cut -c1-140 | head -18; cp $f $d.csv; rm -rf $d; }
  done
done 2>&1 | tee $O/kernel_stats.log
python bench.py 2>$O/bench.err | tee $O/bench.json | cut -c1-600
