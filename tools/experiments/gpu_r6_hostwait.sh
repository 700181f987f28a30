#!/bin/bash
# Host side of a rank: how the waiting thread waits (orbx_set_host_wait) - one rank, and eight ranks sharing one GPU inside 16 cores (the driver's box grants 16).
O=gpurun_out/r06_hostwait; mkdir -p $O
show='import sys,json; r=json.loads([l for l in sys.stdin.read().splitlines() if l.startswith("{")][-1]); print(r["value"], r["ms_per_step"], r["host_cpu"], r.get("latency"))'
(for m in 0 1; do python tools/experiments/host_thread_probe.py $m; done; HSA_ENABLE_INTERRUPT=1 python tools/experiments/host_thread_probe.py 1; HIP_FORCE_QUEUE_PROFILING=0 ROC_SIGNAL_POOL_SIZE=64 python tools/experiments/host_thread_probe.py 1) 2>&1 | tee $O/thread_probe.txt
for m in block spin block spin; do echo -n "8 ranks on one GPU, 16 cores, $m: "; ORBX_BENCH_BACKEND=gloo taskset -c 0-15 python bench.py --gpus 8 --host-wait $m --pairs 32 --steps 20 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-h2d --no-other-configs --no-latency 2>$O/err8.txt | python -c "$show"; done | tee $O/eight_ranks.txt
for m in block spin; do echo -n "2 ranks on one GPU, 4 cores, $m: "; ORBX_BENCH_BACKEND=gloo taskset -c 0-3 python bench.py --gpus 2 --host-wait $m --pairs 64 --steps 40 --warmup 5 --min-seconds 2 --no-cpu-baseline --no-h2d --no-other-configs --no-latency 2>$O/err2.txt | python -c "$show"; done | tee $O/two_ranks.txt
tail -3 $O/err8.txt
