#!/bin/bash
# VALU instruction count per launch of the render kernel for several builds: tools/pmc_valu.sh a.so b.so ...
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for LIB in "$@"; do
  TAG=$(basename $LIB .so)
  OUT=$ROOT/gpurun_out/pmcv_$TAG
  mkdir -p $OUT
  export MI355RT_SO=$ROOT/$LIB
  rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 $ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline --streams 1 > $OUT/log.txt 2>&1 || exit 1
  python3 - <<PY
import csv, glob, collections
f=glob.glob('$OUT/*/*counter_collection.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'render_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
print('$TAG', {k: round(sorted(v)[len(v)//2]) for k,v in sorted(agg.items())})
PY
done
