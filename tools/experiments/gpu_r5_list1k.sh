#!/bin/bash
# round 5, VERDICT r4 item 5 (one more attempt on k_fast_cells): occupancy.  The kernel's LDS (window tile + score tile of the largest cell of the launch + a
# 2-KB survivor / corner list = 7.4 KB per single-wave workgroup at 752 x 480) sets the resident waves per CU; the list at 1 KB (512 entries: the minimum a
# phase-A trip needs) gives 6.4 KB.  A/B of the default line's timed loop, interleaved, same box.  Output: gpurun_out/r05_list1k/ab.txt
O=gpurun_out/r05_list1k
mkdir -p $O
A="--no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic --min-seconds 4"
for rep in 1 2; do
  for v in base ${VARIANTS:-list1k}; do
    if [ $v = base ]; then unset ORBX_BENCH_LIB; else export ORBX_BENCH_LIB=$PWD/build/variants/liborbx_hip_$v.so; fi
    python bench.py $A 2>> $O/err.txt | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', r['value'], r['ms_per_step'], r['parity_check']['identical'], r['stage_ms_alone']['fast_cells'], r['stage_ms_per_step']['fast_cells'], r['config']['library'])" | tee -a $O/ab.txt
  done
done
