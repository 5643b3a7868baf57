#!/bin/bash
# Copy one tools/gpu_session.sh profile session (prof_c2 prof_c2s3 prof_c4 prof_c5 prof_c5spp4) from gpurun_out/ into profiles/:
#   bash tools/collect_profiles.sh <session tag> [round prefix, default r03]
set -e
TAG=$1; R=${2:-r03}
cd "$(dirname "$0")/.."
for wl in c2 c2_streams3 c4 c5 c5spp4; do
  src=gpurun_out/prof_${TAG}_$wl; [ -d $src ] || continue
  mkdir -p profiles/${R}_$wl
  for f in bench.json trace_kernel_stats.csv pmc_summary.json; do [ -f $src/$f ] && cp $src/$f profiles/${R}_$wl/; done
  for f in $src/traffic_${R}*.json $src/valu_${R}*.json; do [ -f $f ] && cp $f profiles/; done
done
sha256sum python-ray-tracer_amd/libmi355rt.so
grep -h so_sha256 profiles/traffic_${R}*.json profiles/valu_${R}*.json | sort | uniq -c
