set -u
export TMPDIR=/tmp
O=gpurun_out/pk1; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "blur or stages or compare or frame420" 2>&1 | tail -3 ) | tee $O/pytest.log
for rep in 1 2; do for v in 0 1; do
  echo "== GZ_BLUR_PK=$v"; GZ_BLUR_PK=$v python tools/run_compare.py 1920 1080 100; GZ_BLUR_PK=$v python tools/run_compare.py 3840 2160 40
done; done 2>&1 | tee $O/ab.log
for sz in "3840 2160 20 4k" "1920 1080 40 1080"; do set -- $sz
  ( cd /tmp && GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/trace$4 -- python $GRAFT_REPO_ROOT/tools/run_compare.py $1 $2 $3 ) > $O/trace$4.log 2>&1
  f=$(find $O/trace$4 -name "*kernel_stats.csv" | head -1)
  echo "== $4 (single stream)"; python3 - $f <<'PY'
import csv,sys
for r in csv.DictReader(open(sys.argv[1])):
    print('%-60s %5s %9.1f'%(r['Name'].replace('gz::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
  cp $f $O/kernel_stats_$4.csv
done
