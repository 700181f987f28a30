#!/bin/bash
# Round 6, second evidence visit: GPU tests of the new rows (blur strips, node pool, 8 ranks on one GPU), the default bench line, facade latency.
O=gpurun_out/r06_b
mkdir -p $O
(cat /sys/fs/cgroup/cpu.max 2>/dev/null; nproc; grep -m1 "model name" /proc/cpuinfo) > $O/host_cpu.txt 2>&1
python -m pytest tests/test_bench_dist.py tests/test_emu_parity.py tests/test_simd_wrappers.py tests/test_multi_comm.py tests/test_matcher_reference.py -x -q -m gpu > $O/pytest_new_rows.log 2>&1; tail -3 $O/pytest_new_rows.log
cp gpurun_out/bench_n8_one_gpu.json $O/ 2>/dev/null
python bench.py > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['bound'], r['roofline']['by_wall'], r['parity_check']['identical'], r.get('value_host_fed'), r.get('h2d_inclusive',{}).get('PCIe_frac'), r.get('dropin_call'), r['latency'], r['host_cpu'], {k: (v.get('value'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()})"; tail -3 $O/bench.err
