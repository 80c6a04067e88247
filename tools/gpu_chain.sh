#!/bin/bash
# Compare-chain iteration: parity of the chain kernels + HIP-event time per chain + per-kernel
# rocprofv3 statistics (single stream) at 4K and 1080p.  Usage: gpu_chain.sh TAG
set -u
export TMPDIR=/tmp
TAG=${1:-c}
O=gpurun_out/$TAG; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or stages or compare or frame420" 2>&1 | tail -3 ) | tee $O/pytest.log
python tools/run_compare.py 3840 2160 30 | tee $O/compare_4k.log
python tools/run_compare.py 1920 1080 60 | tee $O/compare_1080.log
for sz in "3840 2160 20 4k" "1920 1080 40 1080"; do set -- $sz
  ( cd /tmp && GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$4 -- python $GRAFT_REPO_ROOT/tools/run_compare.py $1 $2 $3 ) > $O/trace$4.log 2>&1
  f=$(find $O/trace$4 -name "*kernel_stats.csv" | head -1)
  echo "== $4 (single stream)"; cut -d, -f1-4 $f | sed 's/gz:://g' | cut -c1-130 | head -20
  cp $f $O/kernel_stats_$4.csv
done
