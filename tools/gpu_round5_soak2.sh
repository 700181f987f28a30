#!/bin/bash
# Round 5: second long soak with the final library - the GPU minutes the round has left: more random extractor configurations, more stereo frames / worlds, more
# Kannala-Brandt seeds.  Outputs: gpurun_out/r05_final/soak_xlong_*.txt.
O=gpurun_out/r05_final
mkdir -p $O
timeout 700 python tools/soak_fuzz.py hip 5000 6999 > $O/soak_xlong_fuzz_gpu.txt 2>&1; tail -2 $O/soak_xlong_fuzz_gpu.txt
timeout 500 python tools/soak_reference.py 16 2500 > $O/soak_xlong_vs_reference.txt 2>&1; tail -3 $O/soak_xlong_vs_reference.txt
timeout 500 python tools/soak_round5.py 100 > $O/soak_xlong_round5.txt 2>&1; tail -5 $O/soak_xlong_round5.txt
