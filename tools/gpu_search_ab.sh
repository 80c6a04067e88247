#!/bin/bash
# A/B of builds of the device library on the block search: parity tests of the search with the
# first library in place, then phase A timings (4:4:4 mask 7, 4:2:0 masks 1 and 6) and whole encodes.
# Usage: gpu_search_ab.sh TAG lib1.so lib2.so ...
set -u
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "search or compare_blocks or params or whole_encode" 2>&1 | tail -3 ) | tee $O/pytest.log
{
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib"
  GUETZLI_AMD_LIB=$PWD/$lib python tools/run_search.py 1920 1080
  GUETZLI_AMD_LIB=$PWD/$lib python tools/run_search.py 3840 2160
  GUETZLI_AMD_LIB=$PWD/$lib python tools/run_search.py 1920 1080 1 420
  GUETZLI_AMD_LIB=$PWD/$lib python tools/run_search.py 1920 1080 6 420
done; done
for lib in "$@"; do echo "== encode $lib"; GUETZLI_AMD_LIB=$PWD/$lib python tools/encode_time.py 1920 1080 95 5 | head -2 | cut -c1-400; GUETZLI_AMD_LIB=$PWD/$lib python tools/encode_time.py 3840 2160 95 3 | head -1 | cut -c1-130; done
} 2>&1 | tee $O/ab.log
