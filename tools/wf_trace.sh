#!/bin/bash
# per-kernel durations of the last frame of a workload: bash tools/wf_trace.sh <workload> [steps]
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/wftrace; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $OUT/tr -- python3 $ROOT/bench.py --workload $1 --steps ${2:-6} --warmup 2 --preheat-ms 0 --streams 1 --frames-per-launch 0 --no-cpu-baseline --no-serial --no-host-path --no-dynamic > $OUT/tr.log 2>&1 || { tail -5 $OUT/tr.log; exit 1; }
f=$(find $OUT/tr -name "*kernel_trace.csv" | head -1)
python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for r in rows[-${3:-12}:]:
    print("%-62s %10.1f us  grid %s" % (r["Kernel_Name"][:62], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r.get("Grid_Size")))
PY
rm -rf $OUT/tr
