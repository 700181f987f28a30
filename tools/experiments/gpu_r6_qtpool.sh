#!/bin/bash
# the quadtree's LDS footprint beside FAST: all levels in LDS (45 KB per tree, the product) vs node lists in the global pool (22 KB of tables per tree) for the
# levels above N nodes (ORBX_BENCH_QT_LDS_NODES: 0 = every level in the pool, 150 = levels 0-2, 4000 = none)
O=gpurun_out/r06_qtpool; mkdir -p $O
show='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["parity_check"]["identical"], {k: round(v, 3) for k, v in r["stage_ms_per_step"].items()})'
for rep in 1 2 3; do for n in 4000 0 150; do echo -n "lds_nodes $n: "; ORBX_BENCH_QT_LDS_NODES=$n python bench.py --steps 100 --warmup 10 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic 2>/dev/null | python -c "$show"; done; done | tee $O/qtpool.txt
