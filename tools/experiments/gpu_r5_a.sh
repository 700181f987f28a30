#!/bin/bash
# Round 5, first GPU-box visit: the GPU suite (incl. the headline-shape parity cases of f79dcb5), the default bench line with parity_check,
# a rocprofv3 kernel trace of the bench command.  Outputs under gpurun_out/r05_a/.
O=gpurun_out/r05_a
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; grep -E "passed|failed|error" $O/pytest_gpu.log | tail -3
timeout 600 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc $?"; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r.get('parity_check'), r['cpu_baseline']['value'], {k: v.get('value') for k, v in r['other_configs'].items()}, r['latency'])"; tail -3 $O/bench.err
PROF="--steps 12 --warmup 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --min-seconds 0"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -o trace -- python $R/bench.py $PROF > $R/$O/prof_trace.log 2>&1)
find $O/prof_trace -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats.csv \;
find $O/prof_trace -name '*.csv' -size +4M -delete
head -14 $O/rocprofv3_kernel_stats.csv | cut -c1-160
