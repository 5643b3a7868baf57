#!/bin/bash
# One GPU-box session: tests, the driver's bench command, the default bench, and whatever experiments are listed.
# Usage (via gpurun): bash tools/gpu_session.sh <tag> [steps...]   steps: tests bench20 bench sweep4 sweep5 prof_c2 prof_c2s1 prof_c4 prof_c5 prof_c5spp4 balance hostpath
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $ROOT
for step in "$@"; do
  echo "== $step $(date +%T)"
  case $step in
    tests)    timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/tests.log 2>&1; rc=$?; tail -5 $OUT/tests.log; [ $rc -ne 0 ] && exit $rc ;;
    bench20)  timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench20.err || exit 1; python -c "import json;d=json.load(open('$OUT/bench20.json'));print({k:d[k] for k in ('value','ms_per_step','frame_ms_median','frame_ms_min','launch_ms_median','preheat_ms','host_path_ms')}, d['roofline']['serial'], d.get('cpu_baseline'))" ;;
    bench)    timeout -k 10 300 python bench.py --no-cpu-baseline > $OUT/bench.json 2> $OUT/bench.err || exit 1; python -c "import json;d=json.load(open('$OUT/bench.json'));print({k:d[k] for k in ('value','ms_per_step','frame_ms_median','frame_ms_min','launch_ms_median','rays_traced')})" ;;
    bench_c4|bench_c5|bench_c5spp4)   # the priced bound of the other configs (needs their stamped profiles/valu_r03_*.json of THIS build)
              case $step in bench_c4) WL=c4_3840x2160_s64_d5; ST=192 ;; bench_c5) WL=c5_7680x4320_s256_d8; ST=32 ;; *) WL=c5_7680x4320_s256_d8_spp4; ST=12 ;; esac
              timeout -k 10 300 python bench.py --workload $WL --steps $ST --warmup 4 --no-cpu-baseline --no-serial --no-host-path --no-dynamic > $OUT/$step.json 2> $OUT/$step.err || exit 1
              python -c "import json;d=json.load(open('$OUT/$step.json'));v=d.get('valu') or {};print('$WL', d['ms_per_step'], {k:v.get(k) for k in ('issue_bound_ms','issue_frac','issue_estimate_ms','issue_estimate_frac','salu_bound_ms','lane_utilisation')}, d['roofline'].get('traffic'))" ;;
    collect)  bash tools/collect_profiles.sh $TAG > $OUT/collect.log 2>&1 || exit 1; tail -2 $OUT/collect.log ;;   # this session's stamps into the box's profiles/ (for the bench_* steps after it)
    bench_noev) timeout -k 10 300 python bench.py --no-cpu-baseline --no-step-events --no-serial --no-host-path > $OUT/bench_noev.json 2> $OUT/bench_noev.err || exit 1; python -c "import json;d=json.load(open('$OUT/bench_noev.json'));print(d['ms_per_step'])" ;;
    sweep4)   timeout -k 10 300 python tools/depth_sweep.py --workload c4_3840x2160_s64_d5 > $OUT/sweep4.log 2>&1 || exit 1; cat $OUT/sweep4.log ;;
    sweep5)   timeout -k 10 400 python tools/depth_sweep.py --workload c5_7680x4320_s256_d8 --launches 3 > $OUT/sweep5.log 2>&1 || exit 1; cat $OUT/sweep5.log ;;
    prof_c2)  timeout -k 10 900 python tools/profile_round.py --tag ${TAG}_c2 --pmc-steps 320 --stamp || exit 1 ;;   # (320 steps: the counters' per-frame means then weigh the 16-frame launches as a real run does, not the two single measuring frames)
    prof_c2s3) timeout -k 10 400 python tools/profile_round.py --tag ${TAG}_c2_streams3 --streams 3 --no-pmc || exit 1 ;;
    prof_c4)  timeout -k 10 600 python tools/profile_round.py --tag ${TAG}_c4 --workload c4_3840x2160_s64_d5 --trace-steps 192 --trace-warmup 16 --pmc-steps 32 --sets sq1,sq2,sq3,sq4,fetch,write --stamp || exit 1 ;;
    prof_c5)  timeout -k 10 600 python tools/profile_round.py --tag ${TAG}_c5 --workload c5_7680x4320_s256_d8 --trace-steps 32 --trace-warmup 4 --pmc-steps 8 --sets sq1,sq2,sq3,sq4,fetch,write --stamp || exit 1 ;;
    prof_c5spp4) timeout -k 10 900 python tools/profile_round.py --tag ${TAG}_c5spp4 --workload c5_7680x4320_s256_d8_spp4 --trace-steps 12 --trace-warmup 2 --pmc-steps 4 --sets sq1,sq2,sq3,fetch,write --stamp || exit 1 ;;
    trace_c4) timeout -k 10 300 python tools/profile_round.py --tag ${TAG}_c4 --workload c4_3840x2160_s64_d5 --trace-steps 200 --trace-warmup 10 --no-pmc || exit 1 ;;
    trace_c5) timeout -k 10 300 python tools/profile_round.py --tag ${TAG}_c5 --workload c5_7680x4320_s256_d8 --trace-steps 30 --trace-warmup 3 --no-pmc || exit 1 ;;
    trace_c5spp4) timeout -k 10 400 python tools/profile_round.py --tag ${TAG}_c5spp4 --workload c5_7680x4320_s256_d8_spp4 --trace-steps 12 --trace-warmup 2 --no-pmc || exit 1 ;;
    hostpath) timeout -k 10 300 python tools/host_path_rate.py > $OUT/hostpath.log 2>&1 || exit 1; cat $OUT/hostpath.log ;;
    ab4)      timeout -k 10 300 python tools/ab_bench.py python-ray-tracer_amd/libmi355rt.so python-ray-tracer_amd/libmi355rt.so:64 --workload c4_3840x2160_s64_d5 --rounds 8 --launches 10 > $OUT/ab4.log 2>&1 || exit 1; cat $OUT/ab4.log ;;
    ab5)      timeout -k 10 400 python tools/ab_bench.py python-ray-tracer_amd/libmi355rt.so python-ray-tracer_amd/libmi355rt.so:64 --workload c5_7680x4320_s256_d8 --rounds 5 --launches 3 > $OUT/ab5.log 2>&1 || exit 1; cat $OUT/ab5.log ;;
    thresholds) : > $OUT/thresholds.txt; for wl in x_1920x1080_s16_d3 x_1920x1080_s25_d3 x_1920x1080_s36_d3 x_3840x2160_s36_d5 x_1920x1080_s49_d3 x_3840x2160_s49_d5 c4_3840x2160_s64_d5 x_3840x2160_s100_d5 x_3840x2160_s144_d5 x_3840x2160_s169_d5 x_1920x1080_s196_d3 x_7680x4320_s196_d8 x_3840x2160_s256_d5 c5_7680x4320_s256_d8; do for env in "" "MI355RT_LANES_MINS=97"; do echo "== $wl  [$env]  selection under that environment (none = default; BND_* = bundle pre-cull wherever it can run; LANES_MINS=97 = lane-owned traversal for every clustered scene) vs RT_FLAG_NO_BUNDLES (plain wave-uniform cull)" >> $OUT/thresholds.txt; env $env timeout -k 10 200 python tools/ab_bench.py python-ray-tracer_amd/libmi355rt.so python-ray-tracer_amd/libmi355rt.so:64 --workload $wl --rounds 5 --launches 5 2>&1 | grep -E "median|identical" | sed 's/"sha_u8.*"vs_first"/"vs_first"/' >> $OUT/thresholds.txt || exit 1; done; done; cat $OUT/thresholds.txt ;;
    balance)  timeout -k 10 400 python tools/slab_balance.py > $OUT/balance.log 2>&1 || exit 1; cat $OUT/balance.log ;;
    trace_aa) mkdir -p $OUT/trace_aa && cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_aa -- python3 $ROOT/examples/render_png.py --size 1000x1000 --depth 4 --aa --frames 200 --out $OUT/aa.png > $OUT/trace_aa.log 2>&1 || exit 1; cd $ROOT; cp $(find $OUT/trace_aa -name "*kernel_stats.csv" | head -1) $OUT/aa_kernel_stats.csv; rm -rf $OUT/trace_aa; tail -1 $OUT/trace_aa.log; head -4 $OUT/aa_kernel_stats.csv ;;
    aa)       timeout -k 10 200 python examples/render_png.py --size 1000x1000 --depth 4 --aa --frames 200 --out $OUT/aa.png > $OUT/aa.log 2>&1 || exit 1; cat $OUT/aa.log ;;
    fuzz)     timeout -k 10 400 python tools/fuzz_parity.py --seconds 240 --seed ${FUZZ_SEED:-201} > $OUT/fuzz.log 2>&1; rc=$?; tail -3 $OUT/fuzz.log; [ $rc -ne 0 ] && exit $rc ;;
    *) echo "unknown step $step"; exit 2 ;;
  esac
done
echo "== done $(date +%T)"
