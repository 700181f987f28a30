#!/bin/bash
mkdir -p gpurun_out/r02
python -m pytest tests -x -q -m gpu > gpurun_out/r02/pytest_gpu_d.log 2>&1; tail -3 gpurun_out/r02/pytest_gpu_d.log
python tools/bench_next_rows.py > gpurun_out/r02/next_rows.json 2> gpurun_out/r02/next_rows.err; tail -3 gpurun_out/r02/next_rows.err; grep -A3 "SearchLocalPoints" gpurun_out/r02/next_rows.json
python bench.py --steps 30 --warmup 5 --no-cpu-baseline > gpurun_out/r02/bench_d.json 2> gpurun_out/r02/bench_d.err; cat gpurun_out/r02/bench_d.json | cut -c1-300
