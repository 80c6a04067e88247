#!/bin/bash
# Code size, register and LDS use of every kernel in libguetzli_amd.so (from the gfx950 code
# object's symbol table and kernel descriptors' metadata).  Usage: tools/kernel_sizes.sh [out.csv]
set -e
LIB=$(dirname $0)/../guetzli_amd/libguetzli_amd.so
BIN=/opt/rocm/lib/llvm/bin
T=$(mktemp -d)
$BIN/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950:xnack- --input=<(objcopy -O binary --only-section=.hip_fatbin $LIB /dev/stdout) --output=$T/co.o --unbundle 2>/dev/null || \
  { objcopy -O binary --only-section=.hip_fatbin $LIB $T/fat.bin; $BIN/clang-offload-bundler --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950:xnack- --input=$T/fat.bin --output=$T/co.o --unbundle; }
echo "kernel,code_bytes,vgpr,agpr,sgpr,lds_bytes,scratch_bytes"
$BIN/llvm-readelf --notes $T/co.o > $T/notes.txt
$BIN/llvm-readelf -sW $T/co.o | awk '$4=="FUNC"{print $3","$8}' | sort -t, -k2 > $T/sizes.txt
python3 - $T/notes.txt $T/sizes.txt <<'PY'
import re, sys, subprocess
notes = open(sys.argv[1]).read()
sizes = {}
for l in open(sys.argv[2]):
    sz, name = l.strip().split(",", 1)
    sizes[name] = int(sz)
rows = []
for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)(?=\n\s+- \.a|\namdhsa\.target|\Z)", notes, re.S):
    pass
# kernels are listed as YAML maps; parse fields loosely
for blk in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
    blk = ".agpr_count:" + blk
    def f(k):
        mm = re.search(r"\.%s:\s+(\S+)" % k, blk)
        return mm.group(1) if mm else ""
    name = f("name")
    try:
        dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    except Exception:
        dem = name
    dem = dem.replace("gz::", "")
    rows.append((sizes.get(name, 0), dem[:110], f("vgpr_count"), f("agpr_count"), f("sgpr_count"), f("group_segment_fixed_size"), f("private_segment_fixed_size")))
for sz, dem, v, a, s_, lds, scr in sorted(rows, reverse=True):
    print(f'"{dem}",{sz},{v},{a},{s_},{lds},{scr}')
PY
rm -rf $T
