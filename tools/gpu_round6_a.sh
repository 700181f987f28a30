#!/bin/bash
# Round 6, first GPU visit: the quadtree's node-pool form (mono-init extractors) on the device + the unchanged headline.
O=gpurun_out/r06_a
mkdir -p $O
python -m pytest tests/test_mono_init.py -x -q -m gpu > $O/pytest_mono_init.log 2>&1; tail -3 $O/pytest_mono_init.log
python tools/time_mono_init.py > $O/mono_init_latency.json 2> $O/mono_init_latency.err; cat $O/mono_init_latency.json | head -80; tail -3 $O/mono_init_latency.err
python -m pytest tests -x -q -m gpu --deselect tests/test_mono_init.py > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --no-other-configs > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r['parity_check']['identical'], r.get('h2d_inclusive',{}).get('value'), r['latency'])"; tail -3 $O/bench.err
