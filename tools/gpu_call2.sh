#!/bin/bash
# round 2, GPU visit 2: new FAST kernel - parity, stage times, bench, SQ counters
mkdir -p gpurun_out/r02
python -m pytest tests -x -q -m gpu > gpurun_out/r02/pytest_gpu_a.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu_a.log
python tests/gpu_quick.py > gpurun_out/r02/quick_a.log 2>&1; grep -E "PARITY|^B |DIFF" gpurun_out/r02/quick_a.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_a.json 2> gpurun_out/r02/bench_a.err; cat gpurun_out/r02/bench_a.json | cut -c1-400; tail -2 gpurun_out/r02/bench_a.err
bash tools/gpu_pmc_round.sh > gpurun_out/r02/sq_a.txt 2>&1; grep "^k_" gpurun_out/r02/sq_a.txt
