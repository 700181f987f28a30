#!/bin/bash
# the host-fed line (--h2d: every batch uploaded from pinned memory inside the timed region) against handles in flight and pairs per step
O=gpurun_out/r06_h2d; mkdir -p $O
show='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["parity_check"])'
for shape in "128 4" "128 5" "128 6" "128 8" "64 8" "256 4" "128 4"; do set -- $shape; echo -n "h2d pairs $1 handles $2: "; python bench.py --h2d --pairs $1 --handles $2 --steps 60 --warmup 10 --min-seconds 3 --no-cpu-baseline --no-other-configs --no-latency --no-live-traffic 2>/dev/null | python -c "$show"; done | tee $O/h2d.txt
