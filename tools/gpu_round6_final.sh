#!/bin/bash
# Round 6, the evidence run of the final tree: GPU tests, smoke, the default bench line, serial stage times, mono-init latency, next rows, soaks against the
# reference builds, then rocprofv3 kernel trace + PMC passes (tools/gpu_round6_prof.sh).  Outputs under gpurun_out/r06_final/ (copied to profiles/r06_final by hand).
O=gpurun_out/r06_final
mkdir -p $O
R=$PWD
(cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; grep -m1 "model name" /proc/cpuinfo) > $O/host_cpu.txt 2>&1
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
cp gpurun_out/bench_n8_one_gpu.json $O/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['repeats'], r['timed_seconds'], r['roofline']['frac'], r['roofline']['bound'], r['roofline']['alone_launch_ms'], r['parity_check']['identical'], r.get('value_host_fed'), r.get('h2d_inclusive',{}).get('PCIe_frac'), {k: r['cpu_baseline'][k]['value'] for k in ('one_core','two_cores','all_cores')}, r['cpu_baseline']['all_cores'].get('cores'), {k: (v.get('value'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()}, r['latency'], r.get('dropin_call'), r['host_cpu'])"; tail -3 $O/bench.err
python bench.py --steps 20 --warmup 5 --no-h2d --no-cpu-baseline --no-other-configs > $O/bench_20steps.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_20steps.json')); print('20 steps:', r['value'], r['repeats'], r['block_values']['min'], r['block_values']['max'])"
python bench.py --allgather --steps 100 --warmup 10 --min-seconds 2 --no-cpu-baseline --no-other-configs 2>> $O/bench.err | grep "^{" > $O/bench_allgather_n1.json; python -c "import json; r=json.load(open('$O/bench_allgather_n1.json')); print('allgather', r['value'], r['allgather'])"
python tests/gpu_quick.py > $O/serial_stage_times_and_parity.log 2>&1; grep -E "PARITY|^B |single" $O/serial_stage_times_and_parity.log
python tools/time_mono_init.py > $O/mono_init_latency.json 2> $O/mono_init_latency.err; python -c "import json; r=json.load(open('$O/mono_init_latency.json')); print({k: (v['nFeatures_ms'], v['5x_nFeatures_ms'], v['5x_pool_levels']) for k, v in r.items()})"
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err; tail -2 $O/next_rows.err
python tools/soak_round5.py 12 > $O/soak_round5.txt 2>&1; tail -4 $O/soak_round5.txt
python tools/soak_reference.py 4 200 > $O/soak_vs_reference.txt 2>&1; tail -2 $O/soak_vs_reference.txt
python tools/soak_round3.py 6 > $O/soak_round3.txt 2>&1; tail -2 $O/soak_round3.txt
python tools/soak_round4.py 4 > $O/soak_round4.txt 2>&1; tail -2 $O/soak_round4.txt
python tools/soak_fuzz.py hip 9000 9399 > $O/soak_fuzz_gpu.txt 2>&1; tail -2 $O/soak_fuzz_gpu.txt
bash tools/gpu_round6_prof.sh
