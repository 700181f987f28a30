#!/bin/bash
mkdir -p gpurun_out/r02
for nh in 2 3 4 5 6; do for p in 128; do python bench.py --handles $nh --pairs $p --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('handles $nh pairs $p', r['value'], r['ms_per_step'], r['step_ms'])"; done; done
for p in 64 192 256; do python bench.py --handles 3 --pairs $p --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('handles 3 pairs $p', r['value'], r['ms_per_step'])"; done
GPU_MAX_HW_QUEUES=8 python bench.py --handles 4 --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read()); print('handles 4 hwq8', r['value'], r['ms_per_step'])"
