#!/bin/bash
# Round 6: long soak of the final library with new seeds - random extractor configurations with the quadtree's node pool forced on a varying share of the
# levels, plain random configurations, stereo frames / worlds against the reference builds, the Kannala-Brandt paths.  Outputs: gpurun_out/r06_soak/.
O=gpurun_out/r06_soak
mkdir -p $O
timeout 900 python tools/soak_fuzz.py hip 30000 31999 pool > $O/soak_long_fuzz_node_pool_gpu.txt 2>&1; tail -1 $O/soak_long_fuzz_node_pool_gpu.txt
timeout 700 python tools/soak_fuzz.py hip 32000 33499 > $O/soak_long_fuzz_gpu.txt 2>&1; tail -1 $O/soak_long_fuzz_gpu.txt
timeout 500 python tools/soak_reference.py 16 2000 > $O/soak_long_vs_reference.txt 2>&1; tail -3 $O/soak_long_vs_reference.txt
timeout 500 python tools/soak_round5.py 80 > $O/soak_long_round5.txt 2>&1; tail -5 $O/soak_long_round5.txt
timeout 300 python tools/soak_round4.py 12 > $O/soak_long_round4.txt 2>&1; tail -2 $O/soak_long_round4.txt
timeout 300 python tools/soak_round3.py 12 > $O/soak_long_round3.txt 2>&1; tail -2 $O/soak_long_round3.txt
