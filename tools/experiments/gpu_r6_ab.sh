#!/bin/bash
# A/B of library variants (build/variants/liborbx_hip_<name>.so, tools/experiments/build_r6_variants.sh): the bench line of each, interleaved, REPS times.
# usage: gpu_r6_ab.sh OUTDIR REPS name1 name2 ...
O=gpurun_out/$1; REPS=$2; shift 2
mkdir -p $O
for rep in $(seq 1 $REPS); do
  for v in "$@"; do
    echo -n "$v "
    ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic 2>>$O/err.txt \
      | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['parity_check']['identical'], {k: round(v, 3) for k, v in r['stage_ms_per_step'].items()})"
  done
done | tee $O/ab_bench.txt
