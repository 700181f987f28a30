#!/bin/bash
# bench line (corner_field and natural) for every build/variants/liborbx_hip_base_*.so, twice
O=gpurun_out/sweep
mkdir -p $O
for rep in 1 2; do for f in build/variants/liborbx_hip_base_*.so; do
  n=$(basename $f .so); n=${n#liborbx_hip_base_}
  echo -n "$n "; ORBX_BENCH_LIB=$f python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], {k: round(v, 4) for k, v in r['stage_ms_alone'].items() if k in ('fast_cells',)})"
done; done | tee $O/sweep.txt
for f in build/variants/liborbx_hip_base_*.so; do
  n=$(basename $f .so); n=${n#liborbx_hip_base_}
  echo -n "natural $n "; ORBX_BENCH_LIB=$f python bench.py --workload natural --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'])"
done | tee $O/sweep_natural.txt
