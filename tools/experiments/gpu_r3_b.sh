#!/bin/bash
# A/B visit of the two-run FAST kernel (build/variants/liborbx_hip_prev.so = before, _new = after, base_list* = after with a smaller list):
# GPU parity, k_fast_cells alone, bench lines, natural workload; handle count inside a torch.distributed process
O=gpurun_out/r3b
mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | tee $O/parity.txt
python tools/experiments/time_fast_variants.py prev new base_list 2>&1 | grep "B=" | tee $O/ab_fast.txt
line() { python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], {k: round(v, 4) for k, v in r['stage_ms_alone'].items()})"; }
for v in prev new prev new base_list1k base_list1536; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | line; done | tee $O/ab_bench.txt
for v in prev new; do echo -n "natural $v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --workload natural --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | line; done | tee $O/ab_natural.txt
for nh in 3 4 3 4; do echo -n "dist handles $nh "; ORBX_BENCH_FORCE_DIST=1 python bench.py --handles $nh --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | line; done | tee $O/dist_handles.txt
python tests/gpu_quick.py 2>&1 | grep -E "PARITY|^B |single" | tee $O/quick.txt
