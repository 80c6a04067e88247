set -u
export TMPDIR=/tmp
O=gpurun_out/search4; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search or compare or stages or params or whole_encode" 2>&1 | tail -3 ) | tee $O/pytest.log
for lib in "$@"; do
  ( cd /tmp && GUETZLI_AMD_LIB=$GRAFT_REPO_ROOT/tools/variants/$lib.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/$lib -- python $GRAFT_REPO_ROOT/tools/run_search.py 1920 1080 ) > $O/$lib.log 2>&1
  f=$(find $O/$lib -name "*kernel_stats.csv" | head -1)
  echo "== $lib"; python3 - $f <<'PY'
import csv,sys
for r in list(csv.DictReader(open(sys.argv[1])))[:3]:
    print('%-60s %5s %9.1f us'%(r['Name'].replace('gz::','')[:60], r['Calls'], float(r['AverageNs'])/1000))
PY
  rm -rf $O/$lib
done
for rep in 1 2; do for lib in "$@"; do echo "== $lib"; GUETZLI_AMD_LIB=$PWD/tools/variants/$lib.so python tools/run_compare.py 1920 1080 100; GUETZLI_AMD_LIB=$PWD/tools/variants/$lib.so python tools/run_compare.py 3840 2160 40; done; done
