set -u
export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
cp .gpurun_head $O/head.txt 2>/dev/null || true
( GZ_SINGLE_STREAM=1 timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -25 ) | tee $O/gputests_one_stream.log
( GZ_SINGLE_STREAM=0 timeout 900 python -m pytest tests -m gpu -x -q -n 2 2>&1 | tail -25 ) | tee $O/gputests_three_streams.log
