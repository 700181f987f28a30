#!/bin/bash
# round 2, GPU visit 1: VALU issue-rate survey, stereo tie tests, soak vs the reference builds
mkdir -p gpurun_out/r02
tools/bin/valu_survey > gpurun_out/r02/valu_survey.txt 2>&1; tail -5 gpurun_out/r02/valu_survey.txt
python -m pytest tests/test_stereo_ties.py tests/test_frame_reference.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r02/pytest_ties.log 2>&1; tail -3 gpurun_out/r02/pytest_ties.log
python tools/soak_reference.py 4 1000 > gpurun_out/r02/soak_vs_reference.txt 2>&1; tail -3 gpurun_out/r02/soak_vs_reference.txt
