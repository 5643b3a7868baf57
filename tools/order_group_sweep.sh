#!/bin/bash
# Dispatch-group size sweep (MI355RT_ORDER_GROUP = log2 blocks per XCD-affine group; 0 = every block its own group):
# frame time per workload and launch mode, then WRITE_SIZE per frame of the headline workload.  Run on the GPU box:
#   bash tools/order_group_sweep.sh <tag>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; cd $ROOT
LOG=$OUT/order_group_sweep.txt; : > $LOG
bench() { # workload group extra-args...
  wl=$1; g=$2; shift 2
  MI355RT_ORDER_GROUP=$g timeout -k 10 200 python bench.py --workload $wl --no-cpu-baseline --no-serial --no-host-path --no-dynamic "$@" > $OUT/og.json 2> $OUT/og.err || { tail -3 $OUT/og.err; exit 1; }
  python -c "import json;d=json.load(open('$OUT/og.json'));print('$wl group=$g $*', d['ms_per_step'])" | tee -a $LOG
}
for g in 0 2 3 4 5; do bench c2_1920x1080_s8_d3 $g --steps 960 --streams 1 --frames-per-launch 16; done
for g in 0 2 3 4 5; do bench c2_1920x1080_s8_d3 $g --steps 600 --streams 1 --frames-per-launch 0; done
for g in 0 3 4; do bench c2_1920x1080_s8_d3 $g --steps 900 --streams 3 --frames-per-launch 0; done
for g in 0 2 3 4; do bench x_1920x1080_s25_d3 $g --steps 480 --streams 1 --frames-per-launch 16; done
for g in 0 1 2 3; do bench x_1920x1080_s36_d3 $g --steps 480 --streams 1 --frames-per-launch 16; done
for g in 0 1 2 3; do bench c4_3840x2160_s64_d5 $g --steps 128 --warmup 16 --streams 1 --frames-per-launch 16; done
for g in 0 1 2 3; do bench c4_3840x2160_s64_d5 $g --steps 100 --warmup 10 --streams 1 --frames-per-launch 0; done
cd /tmp && export TMPDIR=/tmp
for mode in "1 16" "3 0"; do set -- $mode; ST=$1; FPL=$2; DIV=$(( FPL > 0 ? FPL : 1 ))
for g in 0 2 3 4 5; do
  export MI355RT_ORDER_GROUP=$g
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/ws_tmp -- python3 $ROOT/bench.py --steps 64 --warmup 16 --preheat-ms 0 --no-serial --no-host-path --no-cpu-baseline --no-dynamic --streams $ST --frames-per-launch $FPL > $OUT/ws_tmp.log 2>&1 || { tail -5 $OUT/ws_tmp.log; exit 1; }
  python3 - <<PY | tee -a $LOG
import csv,glob
f=glob.glob("$OUT/ws_tmp/**/*counter_collection.csv",recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
v=[x for x in v if x > 0.6*max(v)] if $FPL else v      # whole launches only (the first frames of a geometry go one by one)
print("group=$g streams $ST frames/launch $FPL: WRITE_SIZE KB per frame: mean %.0f  last %.0f  (%d launches)" % (sum(v)/len(v)/$DIV, v[-1]/$DIV, len(v)), flush=True)
PY
  rm -rf $OUT/ws_tmp
done; done
