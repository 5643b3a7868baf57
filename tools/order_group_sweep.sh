# Dispatch-group size sweep (MI355RT_ORDER_GROUP = log2 blocks per XCD-affine group; 0 = block-level order):
# frame time per workload and setting, and WRITE_SIZE of the headline frame.  Run on the GPU box.
L=python-ray-tracer_amd/libmi355rt.so
run() { # workload rounds launches extra-args...
  wl=$1; r=$2; n=$3; shift 3
  for g in $GS; do
    printf "%s group=%s %s " "$wl" "$g" "$*"
    MI355RT_ORDER_GROUP=$g python tools/ab_bench.py $L --workload $wl --rounds $r --launches $n "$@" 2>&1 | grep median | sed 's/.*"median_ms": \([0-9.]*\).*/\1/' || exit 1
  done
}
GS="0 2 3 4 5" run c2_1920x1080_s8_d3 12 60 --streams 3
GS="0 3 4 5" run c2_1920x1080_s8_d3 12 30
GS="0 4" run c2_1920x1080_s8_d3 8 20 --aa
GS="0 3 4" run x_1920x1080_s25_d3 8 30 --streams 3
GS="0 2 3" run x_1920x1080_s36_d3 8 30 --streams 3
GS="0 2 3" run c4_3840x2160_s64_d5 5 8
for g in 0 3 4 5; do printf "group=%s " $g; MI355RT_ORDER_GROUP=$g bash tools/write_size.sh $L:3 | grep WRITE_SIZE; done
