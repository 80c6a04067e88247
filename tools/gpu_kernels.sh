#!/bin/bash
# Kernel iteration session: Compare-chain parity (gpu tests of the kernels) + per-kernel
# rocprofv3 stats at 4K and 1080p.  Usage: gpu_kernels.sh TAG [pytest -k expr]
set -u
export TMPDIR=/tmp
TAG=${1:-k}; KEXPR=${2:-"blur or stages or compare or diffmap or encode_matches"}
O=gpurun_out/$TAG; mkdir -p $O
tools/ubench/bw 2>/dev/null | head -3 | tee $O/bw.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "$KEXPR" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log ); tail -4 $O/pytest.log
python tools/run_compare.py 3840 2160 30 | tee $O/compare_4k.log
python tools/run_compare.py 1920 1080 60 | tee $O/compare_1080.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace4k -- python $GRAFT_REPO_ROOT/tools/run_compare.py 3840 2160 20 ) > $O/trace4k.log 2>&1; tail -1 $O/trace4k.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace1080 -- python $GRAFT_REPO_ROOT/tools/run_compare.py 1920 1080 40 ) > $O/trace1080.log 2>&1; tail -1 $O/trace1080.log
for f in $(find $O -name "*kernel_stats.csv"); do echo "== $f"; cut -d, -f1-4 $f | sed 's/gz:://g' | cut -c1-150 | head -24; done
