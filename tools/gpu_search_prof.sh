export TMPDIR=/tmp
O=gpurun_out/r4g; mkdir -p $O
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
for v in s3 s4; do
  cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd.so
  for sz in "1920 1080" "3840 2160"; do
    d=$O/trace_${v}_$(echo $sz | cut -d' ' -f1)
    ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/tools/run_search.py $sz ) > $d.log 2>&1
    f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; }
    echo "== $v $sz"; grep -i "search\|rank" $d.csv | cut -d, -f1-4 | sed 's/gz:://g' | cut -c1-100
  done
  python tools/encode_time.py 3840 2160 95 x 4 | tail -1 | tr ',' '\n' | grep -i "block_search\|'total'"
done
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
