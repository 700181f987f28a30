#!/bin/bash
# round 4: where the fisheye configuration (BASELINE.json configs[2]) spends its step: stage times + rocprofv3 kernel trace
O=gpurun_out/r04_fisheye
mkdir -p $O
R=$PWD
python bench.py --config fisheye --steps 40 --warmup 6 --min-seconds 2 --no-cpu-baseline > $O/bench_fisheye.json 2> $O/err.log; python -c "import json; r=json.load(open('$O/bench_fisheye.json')); print('fisheye', r['value'], r['ms_per_step'], r['stage_ms_alone'])"
export TMPDIR=/tmp
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o trace -- python $R/bench.py --config fisheye --steps 12 --warmup 3 --no-cpu-baseline --min-seconds 0 > $R/$O/prof.log 2>&1)
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats_fisheye.csv \;
head -14 $O/rocprofv3_kernel_stats_fisheye.csv | cut -c1-60,200-330
find $O/prof -name '*.csv' -size +2M -delete
