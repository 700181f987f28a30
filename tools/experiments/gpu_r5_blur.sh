#!/bin/bash
# Round 5: the blur on the matrix cores (k_blur_mfma + k_blur_edges) against the vector kernel: GPU parity of both forms, stage times alone, the headline A/B.
O=gpurun_out/r05_blur
mkdir -p $O
python -m pytest tests/test_blur_modes.py tests/test_gpu_headline_shape.py -x -q -m gpu 2>&1 | tail -3
AB="--steps 200 --warmup 20 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency"
for rep in 1 2; do for m in 1 0; do
  ORBX_BLUR_MODE=$m python bench.py $AB > $O/blur${m}_$rep.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/blur${m}_$rep.json')); print('ORBX_BLUR_MODE=$m rep $rep', r['value'], r['ms_per_step'], r['parity_check']['identical'], 'alone', {k: round(v, 3) for k, v in r['stage_ms_alone'].items()}, 'in flight', {k: round(v, 3) for k, v in r['stage_ms_per_step'].items()})"
done; done
for m in 1 0; do
  ORBX_BLUR_MODE=$m python bench.py $AB --workload natural > $O/blur${m}_natural.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/blur${m}_natural.json')); print('natural ORBX_BLUR_MODE=$m', r['value'], r['ms_per_step'], r['parity_check']['identical'])"
done
tail -3 $O/ab.err
