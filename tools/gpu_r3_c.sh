#!/bin/bash
O=gpurun_out/r03c
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py --config rgbd --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_rgbd.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_rgbd.json')); print('rgbd', r['value'], r['ms_per_step'], r['config']['avg_matches_per_unit'], r['stage_ms_per_step'], r['stage_ms_alone'])"
export TMPDIR=/tmp
R=$PWD
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_rgbd -o trace -- python $R/bench.py --config rgbd --steps 12 --warmup 3 --no-cpu-baseline --min-seconds 0 > $R/$O/prof_rgbd.log 2>&1)
find $O/prof_rgbd -name '*kernel_stats.csv' -exec cp {} $O/rgbd_kernel_stats.csv \;
head -14 $O/rgbd_kernel_stats.csv | cut -c1-160
find $O/prof_rgbd -name '*.csv' -size +2M -delete
