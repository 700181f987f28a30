#!/bin/bash
python tools/time_fast_variants.py blur16 blur24 blur32 2>&1 | grep "B="
for v in blur16 blur24 blur32 blur16 blur24 blur32; do echo -n "$v "; ORBX_BENCH_LIB=build/variants/liborbx_hip_$v.so python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | cut -c125-160; done
