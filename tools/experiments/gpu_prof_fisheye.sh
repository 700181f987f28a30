#!/bin/bash
# rocprofv3 kernel statistics of the fisheye configuration for build/variants/liborbx_hip_{prev,new}.so
O=gpurun_out/ab
mkdir -p $O
export TMPDIR=/tmp
R=$PWD
for v in prev new; do
  (cd /tmp && ORBX_BENCH_LIB=$R/build/variants/liborbx_hip_$v.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_fe_$v -o trace -- python $R/bench.py --config fisheye --steps 12 --warmup 3 --no-cpu-baseline --no-h2d --min-seconds 0 > $R/$O/prof_fe_$v.log 2>&1)
  find $O/prof_fe_$v -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_fisheye_$v.csv \;
  echo "== $v"; head -14 $O/kernel_stats_fisheye_$v.csv | cut -d, -f1-4 | cut -c1-150
  rm -rf $O/prof_fe_$v
done
