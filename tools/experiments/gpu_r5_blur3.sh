#!/bin/bash
# Round 5: the matrix-core blur, band-segment sweep: library variants built with -DORBX_BLUR_SEG_BANDS=4 / 16 beside the tree's 8, and the vector form.
O=gpurun_out/r05_blur
mkdir -p $O
AB="--steps 200 --warmup 20 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency"
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py $AB > $O/sweep_$name.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/sweep_$name.json')); print('$name', r['value'], r['ms_per_step'], r['parity_check']['identical'], 'blur alone', r['stage_ms_alone']['blur'], 'in flight', r['stage_ms_per_step']['blur'])"
}
for rep in 1 2; do
run vector_$rep ORBX_BLUR_MODE=1
run seg8_$rep ORBX_BLUR_MODE=0
run seg4_$rep ORBX_BLUR_MODE=0 ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_seg4.so
run seg16_$rep ORBX_BLUR_MODE=0 ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_seg16.so
done
