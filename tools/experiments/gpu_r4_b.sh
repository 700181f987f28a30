#!/bin/bash
# round 4, second GPU visit: the whole GPU suite on the current build, next rows incl. the new batched forms, allgather through the library at N = 1
O=gpurun_out/r04_b
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err; tail -3 $O/next_rows.err; python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_b/next_rows.json"))
for k, v in r.items():
    if "batched on the device" in k or "SearchLocalPoints" in k: print(k[:90], v)
PY
python bench.py --allgather --steps 60 --warmup 10 --min-seconds 2 --no-cpu-baseline --no-other-configs 2>> $O/bench.err | grep "^{" > $O/bench_allgather_n1.json; python -c "import json; r=json.load(open('$O/bench_allgather_n1.json')); print('allgather', r['value'], r['allgather'])"; tail -2 $O/bench.err
python bench.py --config rgbd --steps 40 --warmup 6 --min-seconds 2 --no-cpu-baseline > $O/bench_rgbd.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_rgbd.json')); print('rgbd', r['value'], r['stage_ms_alone'])"
