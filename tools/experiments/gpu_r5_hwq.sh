#!/bin/bash
# round 5: HIP's hardware-queue count (GPU_MAX_HW_QUEUES, default 4) against the default line's timed loop with the final kernels: four handles x three streams are dealt
# onto the queues round robin.  Output: gpurun_out/r05_hwq/ab.txt
O=gpurun_out/r05_hwq
mkdir -p $O
A="--no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic --min-seconds 3"
for q in ${QS:-default 8 12 2 default}; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py $A 2>> $O/err.txt | python -c "
import json,sys
r=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('GPU_MAX_HW_QUEUES=$q', r['value'], r['ms_per_step'], r['parity_check']['identical'], r['stage_ms_per_step'])" | tee -a $O/ab.txt
done
