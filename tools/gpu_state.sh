#!/bin/bash
# Does the box slow down after sustained load?  Alternates the bandwidth micro-benchmark and
# the Compare chain around the GPU test-suite.
export TMPDIR=/tmp
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -8
tools/ubench/bw | head -2; python tools/run_compare.py 1920 1080 60
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|fclk|Power" | head -8
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
tools/ubench/bw | head -2; python tools/run_compare.py 1920 1080 60
rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temp" | head -12
python tools/encode_time.py 1920 1080 | tail -1 | cut -c1-200
python tools/run_compare.py 1920 1080 60; python tools/run_compare.py 3840 2160 30
sleep 15
python tools/run_compare.py 1920 1080 60; tools/ubench/bw | head -2
