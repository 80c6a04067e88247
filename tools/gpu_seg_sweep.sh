export TMPDIR=/tmp
O=gpurun_out/${1:-r4e}; mkdir -p $O
for seg in 48 96 128; do
  echo "== seg $seg"
  GZ_STREAM_SEG=$seg GZ_SINGLE_STREAM=1 python tools/run_compare.py 3840 2160 20
  GZ_STREAM_SEG=$seg python tools/run_compare.py 3840 2160 20
  GZ_STREAM_SEG=$seg GZ_SINGLE_STREAM=1 python tools/run_compare.py 1920 1080 40
done 2>&1 | tee $O/sweep.log
echo "== separate"; GZ_BLUR_STREAM=0 GZ_SINGLE_STREAM=1 python tools/run_compare.py 3840 2160 20 | tee -a $O/sweep.log; GZ_BLUR_STREAM=0 python tools/run_compare.py 3840 2160 20 | tee -a $O/sweep.log
for seg in 96; do
d=$O/trace_$seg
( cd /tmp && GZ_STREAM_SEG=$seg GZ_SINGLE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$d -- python $GRAFT_REPO_ROOT/tools/run_compare.py 3840 2160 20 ) > $d.log 2>&1
f=$(find $d -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && { cp $f $d.csv; rm -rf $d; }
grep stream $d.csv | cut -d, -f1-4 | sed 's/gz:://g' | cut -c1-60,140-
done
