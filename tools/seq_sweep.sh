#!/bin/bash
# ms per frame of the headline bench for several (streams, frames per launch) settings: bash tools/seq_sweep.sh <tag>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; cd $ROOT
: > $OUT/seq_sweep.txt
for cfg in "3 8" "1 8" "1 16" "1 32" "2 8" "2 16" "3 4" "3 16" "3 0" "1 0"; do
  set -- $cfg
  for steps in 960 20; do
    timeout -k 10 200 python bench.py --streams $1 --frames-per-launch $2 --steps $steps --warmup 8 --no-cpu-baseline --no-serial --no-host-path --no-dynamic > $OUT/ss.json 2> $OUT/ss.err || exit 1
    python -c "import json;d=json.load(open('$OUT/ss.json'));print('streams $1 fpl $2 steps $steps', d['ms_per_step'], 'submit', d['host_submit_ms_per_step'], 'launch_ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'], d['roofline']['per_launch']['frac'])" | tee -a $OUT/seq_sweep.txt
  done
done
