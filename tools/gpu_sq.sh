#!/bin/bash
# SQ issue / wait counters of the Compare chain's kernels (single stream, --pmc only), 4K and 1080p.
# Usage: gpu_sq.sh TAG [env assignments passed to the run, e.g. GZ_BLUR_PK=0]
set -u
export TMPDIR=/tmp
O=gpurun_out/${1:-sq}; mkdir -p $O
shift
ENVS=("$@")
R=$GRAFT_REPO_ROOT
( cd /tmp && rocprofv3 -L > $R/$O/counters_avail.txt 2>&1 )
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
B="SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"
C="SQ_WAVES SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH"
for sz in "3840 2160 4k" "1920 1080 1080"; do set -- $sz
  for p in A B C; do
    eval ctrs=\$$p
    ( cd /tmp && env GZ_SINGLE_STREAM=1 "${ENVS[@]}" timeout 300 rocprofv3 --pmc $ctrs --output-format csv -d $R/$O/sq_$3_$p -- python $R/tools/run_compare.py $1 $2 3 ) > $O/sq_$3_$p.log 2>&1
  done
  python tools/pmc_summary.py $O/sq_$3_A $O/sq_$3_B $O/sq_$3_C > $O/chain_sq_$3.csv 2>$O/summary_$3.err
done
find $O -name "*counter_collection.csv" -delete
python3 - $O/chain_sq_4k.csv <<'PY'
import csv,sys,collections
d=collections.defaultdict(dict)
for r in csv.DictReader(open(sys.argv[1])): d[r['kernel']][r['counter']]=float(r['avg_value'])
for k,v in d.items():
    wc=v.get('SQ_WAVE_CYCLES',0); 
    if not wc: continue
    print('%-50s busy %10.0f valu %4.1f%% lds %4.1f%% vmem %4.1f%% waitinst %4.1f%% ldsconf %s'%(k[:50], v.get('SQ_BUSY_CYCLES',0), 100*v.get('SQ_ACTIVE_INST_VALU',0)/wc, 100*v.get('SQ_ACTIVE_INST_LDS',0)/wc, 100*v.get('SQ_ACTIVE_INST_VMEM',0)/wc, 100*v.get('SQ_WAIT_INST_ANY',0)/wc, v.get('SQ_LDS_BANK_CONFLICT')))
PY
