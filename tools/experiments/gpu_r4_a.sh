#!/bin/bash
# round 4, first GPU visit: the GPU parity suite on the Sophus / Eigen-order build + the default bench line (other_configs, 6-s timed region)
O=gpurun_out/r04_a
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -5 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench.err; tail -3 $O/bench.err
python - <<'PY'
import json
r = json.load(open("gpurun_out/r04_a/bench_n1.json"))
print("headline", r["value"], r["ms_per_step"], r["repeats"], r["timed_seconds"], r["roofline"]["kernel"], r["roofline"]["frac"], r["roofline"]["alone_launch_ms"])
print("stage alone", r["stage_ms_alone"])
print("other", {k: (v.get("value"), v.get("wall_seconds_of_child")) for k, v in r.get("other_configs", {}).items()})
print("latency", r.get("latency"), "h2d", r.get("h2d_inclusive", {}).get("value"), "cpu", r["cpu_baseline"]["value"])
PY
