#!/bin/bash
# round 4: A/B of whole-library variants on the headline bench line only (3 alternating repetitions)
O=gpurun_out/r04_ab
mkdir -p $O
for rep in 1 2 3; do for v in "$@"; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 60 --warmup 10 --min-seconds 2 --no-cpu-baseline --no-h2d --no-other-configs --no-latency 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], {k: round(v, 3) for k, v in r['stage_ms_per_step'].items()})"; done; done | tee -a $O/ab2_bench.txt
