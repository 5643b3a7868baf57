#!/bin/bash
# Run on the GPU box (via gpurun): kernel trace + PMC passes of the default bench command.
# Usage: tools/profile.sh <tag>   -> gpurun_out/prof_<tag>/{trace,pmc_*}/...
set -o pipefail
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $ROOT/bench.py --steps 30 --warmup 5 --no-cpu-baseline"
# kernel trace of the DEFAULT bench command (1000 steps): a 30-step run ends before the clocks have ramped and
# shows ~15 % longer kernels, profiler or not
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $ROOT/bench.py --no-cpu-baseline > $OUT/trace.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_sq1 -- $BENCH > $OUT/pmc_sq1.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU --output-format csv -d $OUT/pmc_sq2 -- $BENCH > $OUT/pmc_sq2.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_grbm -- $BENCH > $OUT/pmc_grbm.log 2>&1 || exit 1
echo done
