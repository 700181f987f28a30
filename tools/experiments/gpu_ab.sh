#!/bin/bash
# A/B of two library builds (build/variants/liborbx_hip_prev.so, liborbx_hip_new.so): GPU parity of the new one, serial stage times, the bench line twice each
O=gpurun_out/ab
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
python tools/experiments/time_fast_variants.py prev new 2>&1 | grep "B=" | tee $O/ab_fast.txt
for v in prev new prev new; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], {k: round(v, 4) for k, v in r['stage_ms_alone'].items()})"; done | tee $O/ab_bench.txt
for v in prev new; do echo -n "natural $v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --workload natural --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['stage_ms_alone']['fast_cells'])"; done | tee $O/ab_natural.txt
