#!/bin/bash
# round 4: A/B of library variants build/variants/liborbx_hip_<name>.so given as arguments: serial stage times + the bench line (twice each, alternating)
O=gpurun_out/r04_ab
mkdir -p $O
python tools/experiments/time_fast_variants.py "$@" 2>&1 | grep "B=" | tee -a $O/ab_fast.txt
for rep in 1 2; do for v in "$@"; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 60 --warmup 10 --min-seconds 1.5 --no-cpu-baseline --no-h2d --no-other-configs 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], {k: round(v, 4) for k, v in r['stage_ms_alone'].items()})"; done; done | tee -a $O/ab_bench.txt
