#!/bin/bash
mkdir -p gpurun_out/r02
python tests/gpu_quick.py > gpurun_out/r02/quick_f.log 2>&1; grep -E "PARITY|^B |single|DIFF|rror" gpurun_out/r02/quick_f.log | head -20
python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-h2d 2>/dev/null | cut -c1-250
