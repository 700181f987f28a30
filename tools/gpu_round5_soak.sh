#!/bin/bash
# Round 5: long soak of the parity families with the final library (outputs: gpurun_out/r05_final/soak_long_*.txt, copied to profiles/r05_final by hand).
# Bounded by `timeout` per family: the round's GPU minutes that are left over go here, a family that does not finish reports how far it came.
O=gpurun_out/r05_final
mkdir -p $O
timeout 420 python tools/soak_round5.py 40 > $O/soak_long_round5.txt 2>&1; tail -5 $O/soak_long_round5.txt
timeout 300 python tools/soak_round4.py 16 > $O/soak_long_round4.txt 2>&1; tail -4 $O/soak_long_round4.txt
timeout 300 python tools/soak_round3.py 20 > $O/soak_long_round3.txt 2>&1; tail -5 $O/soak_long_round3.txt
timeout 420 python tools/soak_reference.py 10 1000 > $O/soak_long_vs_reference.txt 2>&1; tail -3 $O/soak_long_vs_reference.txt
timeout 420 python tools/soak_fuzz.py hip 4000 4799 > $O/soak_fuzz_gpu.txt 2>&1; tail -2 $O/soak_fuzz_gpu.txt
