#!/bin/bash
mkdir -p gpurun_out/r02
python tools/time_fast_variants.py > gpurun_out/r02/fast_variants_1.txt 2>&1; cat gpurun_out/r02/fast_variants_1.txt
python -m pytest tests/test_stereo_ties.py -x -q -m gpu 2>&1 | tail -3
