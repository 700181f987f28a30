#!/bin/bash
# round 4: long soak of the parity families with the final library (outputs: gpurun_out/r04_final/soak_long_*.txt)
O=gpurun_out/r04_final
mkdir -p $O
python tools/soak_round4.py 24 > $O/soak_long_round4.txt 2>&1; tail -5 $O/soak_long_round4.txt
python tools/soak_round3.py 30 > $O/soak_long_round3.txt 2>&1; tail -7 $O/soak_long_round3.txt
python tools/soak_reference.py 12 1000 > $O/soak_long_vs_reference.txt 2>&1; tail -3 $O/soak_long_vs_reference.txt
