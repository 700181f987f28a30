#!/bin/bash
# Short GPU visit: parity + single-handle serial stage timings + bench line.
mkdir -p gpurun_out
python tests/gpu_quick.py > gpurun_out/quick.log 2>&1; grep -E "PARITY|DIFF|^B |Error|error" gpurun_out/quick.log | head -20
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/bench.json 2> gpurun_out/bench.err; cut -c1-1500 gpurun_out/bench.json; tail -3 gpurun_out/bench.err
