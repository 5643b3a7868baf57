# WRITE_SIZE per launch for a list of libraries and stream counts: bash tools/write_size.sh lib[:streams] ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "$@"; do
  lib=${spec%%:*}; st=3; case $spec in *:*) st=${spec##*:};; esac
  export MI355RT_SO=$R/$lib
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/ws_tmp -- python3 $R/bench.py --steps 20 --warmup 3 --preheat-ms 0 --no-serial --no-host-path --no-cpu-baseline --streams $st > $R/gpurun_out/ws_tmp.log 2>&1 || { tail -5 $R/gpurun_out/ws_tmp.log; exit 1; }
  python3 - <<PY
import csv,glob
f=glob.glob("$R/gpurun_out/ws_tmp/**/*counter_collection.csv",recursive=True)[0]
v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "render_kernel" in r["Kernel_Name"]]
print("$spec WRITE_SIZE KB/launch mean %.0f  last5 %.0f  (%d launches)" % (sum(v)/len(v), sum(v[-5:])/5, len(v)), flush=True)
PY
  rm -rf $R/gpurun_out/ws_tmp
done
