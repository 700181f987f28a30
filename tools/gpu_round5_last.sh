#!/bin/bash
# round 5, last visit: the GPU suite of the final library (lifetime counters, communicator buffers released) and the default bench line with the
# block count calibrated on 40 warm steps.  Outputs: gpurun_out/r05_last2/
O=gpurun_out/r05_last2
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc $?"
python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['repeats'], r['timed_seconds'], r['roofline']['frac'], r['roofline']['traffic'], r['parity_check']['identical'], {k: (v.get('value'), v.get('timed_seconds'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()})"; tail -3 $O/bench.err
