#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_parity.py tests/test_frame_reference.py tests/test_facade.py -x -q -m gpu > gpurun_out/r02/pytest_gpu_c.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu_c.log
python tests/gpu_quick.py > gpurun_out/r02/quick_c.log 2>&1; grep -E "PARITY|^B |DIFF|single" gpurun_out/r02/quick_c.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_c.json 2> gpurun_out/r02/bench_c.err; cat gpurun_out/r02/bench_c.json | cut -c1-300; tail -2 gpurun_out/r02/bench_c.err
bash tools/gpu_pmc_round.sh > gpurun_out/r02/sq_c.txt 2>&1; grep "^k_" gpurun_out/r02/sq_c.txt
