#!/bin/bash
# round 5, last visit: the GPU suite (with the smallest / largest image cases and the 4600-image batch), smoke(), the default bench line of the final tree, and the
# GPU minutes that are left as soaks with new seeds.  Outputs: gpurun_out/r05_last4/
O=gpurun_out/r05_last4
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc $?"
python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['timed_seconds'], r['roofline']['frac'], r['roofline']['traffic'], r['parity_check']['identical'], {k: (v.get('value'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()})"
timeout 420 python tools/soak_fuzz.py hip 7000 8199 > $O/soak_fuzz_gpu_7000.txt 2>&1; tail -2 $O/soak_fuzz_gpu_7000.txt
timeout 300 python tools/soak_round5.py 60 > $O/soak_round5_60.txt 2>&1; tail -4 $O/soak_round5_60.txt
timeout 300 python tools/soak_reference.py 12 1200 > $O/soak_vs_reference.txt 2>&1; tail -3 $O/soak_vs_reference.txt
