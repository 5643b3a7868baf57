#!/bin/bash
# ms per frame of the headline bench under the driver's command (--steps 20 --warmup 5) and the default one, for several
# frames-per-launch settings, three runs each: bash tools/fpl_sweep.sh <tag>
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$1; mkdir -p $OUT; cd $ROOT
: > $OUT/fpl_sweep.txt
for fpl in 16 20 32 64; do
  for args in "--steps 20 --warmup 5" "--steps 1000 --warmup 50"; do
    for rep in 1 2 3; do
      timeout -k 10 200 python bench.py --frames-per-launch $fpl $args --no-cpu-baseline --no-serial --no-host-path --no-dynamic > $OUT/fs.json 2> $OUT/fs.err || exit 1
      python -c "import json;d=json.load(open('$OUT/fs.json'));print('fpl $fpl $args:', d['ms_per_step'], 'launch_ms', d['roofline']['kernel_ms'], 'frames/launch', d['roofline'].get('frames_per_launch'))" | tee -a $OUT/fpl_sweep.txt
    done
  done
done
