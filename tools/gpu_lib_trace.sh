#!/bin/bash
# Per-kernel statistics (single stream) + chain time of several builds of the device library.
# Usage: gpu_lib_trace.sh TAG lib1.so lib2.so ...
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
LIBS=("$@")
for lib in "${LIBS[@]}"; do n=$(basename $lib .so)
  for sz in "3840 2160 20 4k" "1920 1080 40 1080"; do set -- $sz
  ( cd /tmp && GUETZLI_AMD_LIB=$GRAFT_REPO_ROOT/$lib GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/t$n$4 -- python $GRAFT_REPO_ROOT/tools/run_compare.py $1 $2 $3 ) > $O/t$n$4.log 2>&1
  f=$(find $O/t$n$4 -name "*kernel_stats.csv" | head -1)
  cp $f $O/kernel_stats_${n}_$4.csv; rm -rf $O/t$n$4
  done
done
python3 - $O "${LIBS[@]}" <<'PY'
import csv,sys,os,collections
O=sys.argv[1]; libs=[os.path.basename(l)[:-3] for l in sys.argv[2:]]
for sz in ('4k','1080'):
    tab=collections.OrderedDict()
    for n in libs:
        for r in csv.DictReader(open(f'{O}/kernel_stats_{n}_{sz}.csv')):
            k=r['Name'].replace('gz::','').split('(')[0][:58]
            tab.setdefault(k,{})[n]=float(r['AverageNs'])/1000*int(r['Calls'])/ (20 if sz=='4k' else 40)
    print('==',sz,'us per chain', libs)
    tot={n:0 for n in libs}
    for k,v in tab.items():
        if k.startswith('k_linear') or k.startswith('k_encode') or k.startswith('k_quantize') or 'copyBuffer' in k: continue
        print('%-58s'%k,' '.join('%8.1f'%v.get(n,0) for n in libs))
        for n in libs: tot[n]+=v.get(n,0)
    print('%-58s'%'sum',' '.join('%8.1f'%tot[n] for n in libs))
PY
for rep in 1 2; do for lib in "${LIBS[@]}"; do echo "== $lib"; GUETZLI_AMD_LIB=$PWD/$lib python tools/run_compare.py 1920 1080 100; GUETZLI_AMD_LIB=$PWD/$lib python tools/run_compare.py 3840 2160 40; done; done
