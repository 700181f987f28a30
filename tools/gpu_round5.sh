#!/bin/bash
# One GPU-box visit of round 5 (the final evidence run): gpu tests, the default bench line (headline + other_configs incl. natural imagery + cpu baseline
# + PCIe probe), all-gather through the library at N = 1, serial stage times, next rows (incl. the single-call C ABI rows), soaks against the reference
# (worlds / stereo frames, the round-3 / round-4 batched paths, the round-5 Kannala-Brandt paths), rocprofv3 kernel trace + PMC passes.
# Outputs under gpurun_out/r05_final/ (copied to profiles/r05_final by hand).
O=gpurun_out/r05_final
mkdir -p $O
R=$PWD
(cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us /sys/fs/cgroup/cpu/cpu.cfs_period_us 2>/dev/null; nproc; grep -m1 "model name" /proc/cpuinfo) > $O/host_cpu.txt 2>&1
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['repeats'], r['timed_seconds'], r['roofline']['frac'], r['roofline']['alone_launch_ms'], r['parity_check']['identical'], r.get('h2d_inclusive',{}).get('value'), r.get('h2d_inclusive',{}).get('PCIe_frac'), {k: r['cpu_baseline'][k]['value'] for k in ('one_core','two_cores','all_cores')}, r['cpu_baseline']['all_cores'].get('cores'), {k: (v.get('value'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()}, r['latency'])"; tail -3 $O/bench.err
python bench.py --steps 20 --warmup 5 --no-h2d --no-cpu-baseline --no-other-configs > $O/bench_20steps.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_20steps.json')); print('20 steps:', r['value'], r['repeats'], r['block_values']['min'], r['block_values']['max'])"
python bench.py --allgather --steps 100 --warmup 10 --min-seconds 2 --no-cpu-baseline --no-other-configs 2>> $O/bench.err | grep "^{" > $O/bench_allgather_n1.json; python -c "import json; r=json.load(open('$O/bench_allgather_n1.json')); print('allgather', r['value'], r['allgather'])"
python tests/gpu_quick.py > $O/serial_stage_times_and_parity.log 2>&1; grep -E "PARITY|^B |single" $O/serial_stage_times_and_parity.log
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err; tail -2 $O/next_rows.err; python -c "import json; r=json.load(open('$O/next_rows.json')); print(r.get('single calls at the C ABI, key frames resident (the unchanged facade call sites)'))"
python tools/soak_round5.py 12 > $O/soak_round5.txt 2>&1; tail -5 $O/soak_round5.txt
python tools/soak_reference.py 4 200 > $O/soak_vs_reference.txt 2>&1; tail -2 $O/soak_vs_reference.txt
python tools/soak_round3.py 6 > $O/soak_round3.txt 2>&1; tail -3 $O/soak_round3.txt
python tools/soak_round4.py 4 > $O/soak_round4.txt 2>&1; tail -3 $O/soak_round4.txt
bash tools/gpu_round5_prof.sh
