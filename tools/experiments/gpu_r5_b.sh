#!/bin/bash
# Round 5, second GPU-box visit: the GPU suite on the reference's own camera models (fisheye depths bit-exact), the default bench line.
O=gpurun_out/r05_b
mkdir -p $O
timeout 1200 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc $?"; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('parity_check',{}).get('identical'), r['cpu_baseline']['value'], {k: (v.get('value'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()}, r['latency'])"; tail -3 $O/bench.err
