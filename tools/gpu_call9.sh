#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests/test_kb8.py tests/test_local_points.py -x -q -m gpu 2>&1 | tail -2
python bench.py > gpurun_out/r02/bench_e.json 2> gpurun_out/r02/bench_e.err; python - <<'PY'
import json
r=json.load(open('gpurun_out/r02/bench_e.json'))
print(r['value'], r['ms_per_step'], r['step_ms'], r.get('h2d_inclusive'))
print(r['roofline'])
print(r['stage_ms_alone'], r['reference_stage_ms_alone'])
cb=r.get('cpu_baseline',{}); print({k:cb.get(k) for k in ('value','cores','kind')}, cb.get('one_core',{}).get('value'), cb.get('two_cores',{}).get('value'), cb.get('stage_ms'))
PY
tail -3 gpurun_out/r02/bench_e.err
for c in mono fisheye rgbd; do python bench.py --config $c --steps 40 --warmup 5 > gpurun_out/r02/bench_$c.json 2> gpurun_out/r02/bench_$c.err; python -c "
import json; r=json.load(open('gpurun_out/r02/bench_$c.json')); print('$c', r['value'], r['unit'], r['ms_per_step'], r['roofline']['kernel'], r['config']['avg_matches_per_unit'])"; tail -2 gpurun_out/r02/bench_$c.err; done
