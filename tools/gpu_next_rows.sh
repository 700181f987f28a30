#!/bin/bash
mkdir -p gpurun_out/r02
python tools/bench_next_rows.py > gpurun_out/r02/next_rows_b.json 2> gpurun_out/r02/next_rows_b.err; tail -3 gpurun_out/r02/next_rows_b.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r02/next_rows_b.json'))
for k,v in d.items():
    if isinstance(v,dict): print("%-90s %s"%(k[:90], {a:b for a,b in v.items() if 'ms' in a}))
PY
