#!/bin/bash
# round 5, last visit: the GPU suite and the default bench line of the final tree (lifetime counters, communicator buffers released, RCCL taken from the
# library's own HIP runtime, null-argument checks; device sources unchanged: stamp bcd461438c098f93), one short soak.  Outputs: gpurun_out/r05_last3/
O=gpurun_out/r05_last3
mkdir -p $O
python -m pytest tests -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed" $O/pytest_gpu.log | tail -2
python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc $?"
python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['repeats'], r['timed_seconds'], r['roofline']['frac'], r['roofline']['traffic'], r['roofline']['traffic_source']['stale'], r['parity_check']['identical'], {k: (v.get('value'), v.get('timed_seconds'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()})"; tail -3 $O/bench.err
python tools/soak_round5.py 12 > $O/soak_round5.txt 2>&1; tail -4 $O/soak_round5.txt
