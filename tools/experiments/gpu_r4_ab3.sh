#!/bin/bash
# round 4: A/B of library variants on small batches (single pair, B = 16 / 64): tests/gpu_quick.py stage times + quadtree phases + single-pair loop
O=gpurun_out/r04_ab
mkdir -p $O
for rep in 1 2; do for v in "$@"; do echo "== $v"; ORBX_QUICK_LIB=build/variants/liborbx_hip_$v.so python tests/gpu_quick.py 2>&1 | grep -E "PARITY|^B |quadtree L0|single"; done; done | tee -a $O/ab3_quick.txt
