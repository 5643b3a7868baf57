cd $GRAFT_REPO_ROOT; OUT=gpurun_out/r03e; mkdir -p $OUT
run() { env "$1" timeout -k 10 300 python bench.py --workload $2 --no-cpu-baseline --no-serial --no-host-path --no-dynamic $3 > $OUT/c5.json 2>$OUT/c5.err || { tail -3 $OUT/c5.err; exit 1; }; python -c "import json;d=json.load(open('$OUT/c5.json'));print('$1 $2 $3', d['ms_per_step'])"; }
run A=1 c5_7680x4320_s256_d8 "--steps 24 --warmup 4 --streams 1 --frames-per-launch 0"
run MI355RT_POOL_REPACK=0 c5_7680x4320_s256_d8 "--steps 24 --warmup 4 --streams 1 --frames-per-launch 0"
run MI355RT_POOL=0 c5_7680x4320_s256_d8 "--steps 24 --warmup 4 --streams 1 --frames-per-launch 0"
