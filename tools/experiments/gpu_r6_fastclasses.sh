#!/bin/bash
# k_fast_cells in two launches by cell size class (ORBX_FAST_CLASSES=1): levels whose cells are at most 10 % taller than level 0's get LDS tiles sized for THEM
# (6.4 KB per cell-wave instead of 7.4 KB: 25 instead of 21 cells per CU), the tall cells of the top levels keep the big tiles.  Same library, env switch.
O=gpurun_out/r06_fastclasses; mkdir -p $O
show='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["parity_check"]["identical"], {k: round(v, 3) for k, v in r["stage_ms_per_step"].items()}, r["stage_ms_alone"]["fast_cells"])'
for rep in 1 2 3; do for c in 0 1; do echo -n "classes $c: "; ORBX_FAST_CLASSES=$c python bench.py --steps 100 --warmup 10 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic 2>/dev/null | python -c "$show"; done; done | tee $O/fastclasses.txt
for c in 0 1; do echo -n "natural classes $c: "; ORBX_FAST_CLASSES=$c python bench.py --workload natural --steps 100 --warmup 10 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic 2>/dev/null | python -c "$show"; done | tee -a $O/fastclasses.txt
