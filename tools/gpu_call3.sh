#!/bin/bash
# round 2, GPU visit 3: FAST kernel with flush-based scoring - parity, stage times, bench, SQ counters; LDS byte-read microbenchmark
mkdir -p gpurun_out/r02
tools/bin/lds_u8 > gpurun_out/r02/lds_u8_microbench.txt 2>&1; cat gpurun_out/r02/lds_u8_microbench.txt
python -m pytest tests/test_gpu_parity.py tests/test_frame_reference.py -x -q -m gpu > gpurun_out/r02/pytest_gpu_b.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu_b.log
python tests/gpu_quick.py > gpurun_out/r02/quick_b.log 2>&1; grep -E "PARITY|^B |DIFF" gpurun_out/r02/quick_b.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_b.json 2> gpurun_out/r02/bench_b.err; cat gpurun_out/r02/bench_b.json | cut -c1-300; tail -2 gpurun_out/r02/bench_b.err
bash tools/gpu_pmc_round.sh > gpurun_out/r02/sq_b.txt 2>&1; grep "^k_" gpurun_out/r02/sq_b.txt
