#!/bin/bash
# The descent kernels alone (tools/run_descent.py): per-kernel durations by rocprofv3 --kernel-trace
# for each device-library variant given (guetzli_amd/variants/<name>.so; "-" = the library in the
# tree), and with PMC=1 the SQ / memory counters of the library in the tree.
# Usage: gpurun -- 'bash tools/gpu_descent_prof.sh TAG [N] [variant ...]'
set -u
export TMPDIR=/tmp
TAG=$1; N=${2:-3200000}; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
R=$GRAFT_REPO_ROOT
cp .gpurun_head $O/head.txt 2>/dev/null || true
cp guetzli_amd/libguetzli_amd.so /tmp/lib_orig.so
[ $# = 0 ] && set -- -
for v in "$@"; do
  [ "$v" != "-" ] && cp guetzli_amd/variants/$v.so guetzli_amd/libguetzli_amd.so
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$O/tr_$v -- python $R/tools/run_descent.py $N 12 65536 ${KIND:-random} ) > $O/tr_$v.log 2>&1
  tail -1 $O/tr_$v.log
  python3 - $O/tr_$v <<'PY' | tee $O/levels_$v.txt
import csv, glob, os, sys, collections
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"].split("(")[0].replace("gz::", ""), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows if "k_desc" in r["Kernel_Name"]]
# a descent = the launches between two k_desc_count of level 0: group by position in the descent
per = collections.defaultdict(list)
i = 0
for name, d in seq:
    if name == "k_desc_count" and i and i % 2 == 0 and False: pass
    per[i].append(d); i += 1
# the harness launches 12 levels (24 kernels) per descent
L = 24
med = lambda v: sorted(v)[len(v) // 2]
nd = len(seq) // L
out = []
for j in range(L):
    v = [seq[k * L + j][1] for k in range(1, nd)]   # (skip the first descent: cold)
    if v: out.append("%s %.1f" % ("c" if j % 2 == 0 else "s", med(v)))
print("median us per launch (count/swap by level):", ", ".join(out))
print("sum %.1f us" % sum(float(x.split()[1]) for x in out))
PY
  rm -rf $O/tr_$v
done
cp /tmp/lib_orig.so guetzli_amd/libguetzli_amd.so
if [ "${PMC:-0}" = 1 ]; then
  A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
  B="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES"
  for p in A B; do
    eval ctrs=\$$p
    ( cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/$O/pmc_$p -- python $R/tools/run_descent.py $N 3 65536 ${KIND:-random} ) > $O/pmc_$p.log 2>&1
    tail -2 $O/pmc_$p.log
  done
  python3 - $O <<'PY' | tee $O/pmc_first_level.txt
import csv, glob, os, sys, collections
# counters of the FIRST launch of k_desc_count / k_desc_swap of the last descent of each pass
for p in "AB":
    fs = glob.glob(os.path.join(sys.argv[1], "pmc_" + p, "**", "*counter_collection.csv"), recursive=True)
    if not fs: print("pass", p, "no output"); continue
    rows = list(csv.DictReader(open(fs[0])))
    byk = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0].replace("gz::", "")
        if "k_desc" not in k: continue
        byk[k][r["Counter_Name"]].append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    for k, cs in byk.items():
        for c, v in cs.items():
            v.sort()
            # level 0 of the last descent: launches come in 24s per descent -> the 12th from the end of this kernel's list
            lvl0 = v[-12][1] if len(v) >= 12 else v[0][1]
            print(p, k, c, "level0 %.0f" % lvl0, "level1 %.0f" % (v[-11][1] if len(v) >= 11 else -1))
PY
  find $O -name "*counter_collection.csv" -delete
fi
