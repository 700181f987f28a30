#!/bin/bash
# A/B of two library builds on the fisheye configuration (build/variants/liborbx_hip_prev.so, liborbx_hip_new.so) + the GPU tests that cover the 2-NN
O=gpurun_out/ab
mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_kb8.py tests/test_frame_reference.py -x -q -m gpu 2>&1 | tail -3
for v in prev new prev new; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so timeout 300 python bench.py --config fisheye --steps 60 --warmup 6 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print(r['value'], r['ms_per_step'], {k: round(v, 4) for k, v in r['stage_ms_alone'].items()})"; done | tee $O/ab_fisheye.txt
