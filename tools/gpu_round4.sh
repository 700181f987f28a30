#!/bin/bash
# One GPU-box visit of round 4 (the final evidence run): gpu tests, the default bench line (headline + other_configs + cpu baseline), natural workload,
# all-gather through the library at N = 1, next rows, soaks against the reference, rocprofv3 kernel trace + PMC passes.  Outputs under
# gpurun_out/r04_final/ (copied to profiles/r04_final by hand).
O=gpurun_out/r04_final
mkdir -p $O
R=$PWD
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['repeats'], r['timed_seconds'], r['roofline']['frac'], r['roofline']['alone_launch_ms'], r.get('h2d_inclusive',{}).get('value'), r['cpu_baseline']['value'], {k: v.get('value') for k, v in r['other_configs'].items()}, r['latency'])"; tail -3 $O/bench.err
python bench.py --steps 20 --warmup 5 --no-h2d --no-cpu-baseline --no-other-configs > $O/bench_20steps.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_20steps.json')); print('20 steps:', r['value'], r['repeats'], r['block_values']['min'], r['block_values']['max'])"
python bench.py --workload natural --steps 100 --warmup 10 --min-seconds 2 --no-cpu-baseline --no-h2d --no-other-configs > $O/bench_natural.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_natural.json')); print('natural', r['value'], r['config']['fast_corner_density_t7'], r['roofline']['frac'], r['stage_ms_alone'])"
python bench.py --allgather --steps 100 --warmup 10 --min-seconds 2 --no-cpu-baseline --no-other-configs 2>> $O/bench.err | grep "^{" > $O/bench_allgather_n1.json; python -c "import json; r=json.load(open('$O/bench_allgather_n1.json')); print('allgather', r['value'], r['allgather'])"
python tests/gpu_quick.py > $O/serial_stage_times_and_parity.log 2>&1; grep -E "PARITY|^B |single" $O/serial_stage_times_and_parity.log
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err; tail -2 $O/next_rows.err
python tools/soak_reference.py 4 300 > $O/soak_vs_reference.txt 2>&1; tail -2 $O/soak_vs_reference.txt
python tools/soak_round3.py 12 > $O/soak_round3.txt 2>&1; tail -5 $O/soak_round3.txt
python tools/soak_round4.py 6 > $O/soak_round4.txt 2>&1; tail -5 $O/soak_round4.txt
bash tools/gpu_round4_prof.sh
