#!/bin/bash
O=gpurun_out/r06_shapes2; mkdir -p $O
show='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["parity_check"]["identical"])'
for rep in 1 2 3; do for shape in "128 4" "256 4" "320 4" "384 4"; do set -- $shape; echo -n "pairs $1 handles $2: "; python bench.py --pairs $1 --handles $2 --steps 60 --warmup 10 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic 2>/dev/null | python -c "$show"; done; done | tee $O/shapes.txt
