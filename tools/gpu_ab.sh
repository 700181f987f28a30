#!/bin/bash
mkdir -p gpurun_out/r02
python tools/time_fast_variants.py prev new 2>&1 | grep "B=" | tee gpurun_out/r02/qt_ab.txt
for v in prev new prev new; do ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | cut -c125-200; done
