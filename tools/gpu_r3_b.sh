#!/bin/bash
# round 3, visit B: GPU tests (new: batched local points, rig frustum, PredictScale), bench lines of all configs, natural workload, all-gather at N = 1
O=gpurun_out/r03b
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py --no-h2d > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print(r['value'], r['ms_per_step'], r['repeats'], r['block_values']['min'], r['block_values']['max'], r['roofline']['frac'], r['cpu_baseline']['value'])"
python bench.py --steps 20 --warmup 5 --no-h2d --no-cpu-baseline > $O/bench_20steps.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_20steps.json')); print('20 steps:', r['value'], r['repeats'], r['block_values']['min'], r['block_values']['max'])"
for c in rgbd mono fisheye; do python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_$c.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_$c.json')); print('$c', r['value'], r['ms_per_step'], r['config']['avg_matches_per_unit'], r['stage_ms_per_step'])"; done
python bench.py --workload natural --steps 100 --warmup 10 --no-cpu-baseline --no-h2d > $O/bench_natural.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_natural.json')); print('natural', r['value'], r['config']['fast_corner_density_t7'], r['stage_ms_alone'])"
python bench.py --allgather --steps 100 --warmup 10 --no-cpu-baseline > $O/bench_allgather_n1.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_allgather_n1.json')); print('allgather', r['value'], r['allgather'])"
tail -5 $O/bench.err
