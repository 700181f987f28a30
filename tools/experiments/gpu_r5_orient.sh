#!/bin/bash
# Round 5: k_orient_brief with the patches of all keypoints of a wave requested up front and the next keypoint's BRIEF window prefetched, against the tree before
# (build/variants/liborbx_hip_prev.so): GPU parity, stage times alone, headline A/B, one pair per call.
O=gpurun_out/r05_orient
mkdir -p $O
python -m pytest tests/test_gpu_parity.py tests/test_gpu_headline_shape.py -x -q -m gpu 2>&1 | tail -1
AB="--steps 200 --warmup 20 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency"
run() { name=$1; shift; extra=$1; shift
  env "$@" python bench.py $AB $extra > $O/$name.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/$name.json')); print('$name', r['value'], r['ms_per_step'], r['parity_check']['identical'], 'orient alone', r['stage_ms_alone']['orient_brief'], 'in flight', r['stage_ms_per_step']['orient_brief'])"
}
for rep in 1 2; do
run prev_$rep "" ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_prev.so
run new_$rep ""  X=1
done
run prev_nat "--workload natural" ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_prev.so
run new_nat "--workload natural" X=1
python tools/single_pair_loop.py 300 2>&1 | tail -1
