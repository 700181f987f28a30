#!/bin/bash
# Round 5: the matrix-core blur, panel-width sweep: four tiles per wave (the tree) against two (-DORBX_BLUR_PANEL_TILES=2), and the vector form; corner field and natural imagery.
O=gpurun_out/r05_blur
mkdir -p $O
AB="--steps 200 --warmup 20 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency"
run() { name=$1; shift; extra=$1; shift
  env "$@" python bench.py $AB $extra > $O/sweep_$name.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/sweep_$name.json')); print('$name', r['value'], r['ms_per_step'], r['parity_check']['identical'], 'blur alone', r['stage_ms_alone']['blur'], 'in flight', r['stage_ms_per_step']['blur'])"
}
for rep in 1 2; do
run vector_$rep "" ORBX_BLUR_MODE=1
run t4_$rep "" ORBX_BLUR_MODE=0
run t2_$rep "" ORBX_BLUR_MODE=0 ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_t2.so
done
run vector_nat "--workload natural" ORBX_BLUR_MODE=1
run t4_nat "--workload natural" ORBX_BLUR_MODE=0
run t2_nat "--workload natural" ORBX_BLUR_MODE=0 ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_t2.so
