#!/bin/bash
# round 3, visit A (produced profiles/r03/fast_dense_vs_list_ab.txt): GPU parity of the dense FAST kernel of DESIGN.md section 4c, A/B against the list kernel
# (build/variants/liborbx_hip_prev.so = the list kernel, _new.so = the dense one), both workloads
O=gpurun_out/r03a
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python tools/experiments/time_fast_variants.py prev new 2>&1 | grep "B=" | tee $O/ab_fast.txt
for v in prev new prev new; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['repeats'], r['block_values']['min'], r['block_values']['max'], r['stage_ms_alone'])"; done | tee $O/ab_bench.txt
for v in prev new; do echo -n "natural $v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --workload natural --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], r['config']['fast_corner_density_t7'], r['config']['avg_keypoints_per_image'], r['stage_ms_alone'])"; done | tee $O/ab_natural.txt
python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench.err; cut -c1-200 $O/bench_20.json
