#!/bin/bash
# PMC pass (SQ instruction mix) for one build: tools/pmc.sh <tag> <lib.so>
set -o pipefail
TAG=$1; LIB=$2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p $OUT
export MI355RT_SO=$ROOT/$LIB
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/sq1 -- $BENCH > $OUT/sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU --output-format csv -d $OUT/sq2 -- $BENCH > $OUT/sq2.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_CVT --output-format csv -d $OUT/sq3 -- $BENCH > $OUT/sq3.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_IOPS --output-format csv -d $OUT/sq4 -- $BENCH > $OUT/sq4.log 2>&1 || exit 1
python3 - <<PY
import csv, glob, collections
for d in ['sq1','sq2','sq3','sq4']:
    f=glob.glob('$OUT/'+d+'/*/*counter_collection.csv')[0]
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'render_kernel' in r['Kernel_Name']: agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in sorted(agg.items()): print('$TAG',k, round(sum(v)/len(v)))
PY
