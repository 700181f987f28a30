#!/bin/bash
# One GPU-box visit: gpu tests, bench, rocprof kernel trace + PMC passes.  Outputs under gpurun_out/.
mkdir -p gpurun_out
R=$PWD
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu.log 2>&1; tail -3 gpurun_out/pytest_gpu.log
python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
python tests/gpu_quick.py > gpurun_out/quick.log 2>&1; grep -E "PARITY|^B " gpurun_out/quick.log
python tools/bench_next_rows.py > gpurun_out/next_rows.json 2> gpurun_out/next_rows.err; tail -2 gpurun_out/next_rows.err
python tools/bench_other_configs.py > gpurun_out/other_configs.json 2>/dev/null
export TMPDIR=/tmp
rm -rf gpurun_out/prof_trace gpurun_out/prof_fetch gpurun_out/prof_write
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_trace -o trace -- python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/prof_trace.log 2>&1)
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -o fetch -- python $R/bench.py --steps 4 --warmup 2 --pairs 64 --no-cpu-baseline > $R/gpurun_out/prof_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -o write -- python $R/bench.py --steps 4 --warmup 2 --pairs 64 --no-cpu-baseline > $R/gpurun_out/prof_write.log 2>&1)
python tools/pmc_summary.py gpurun_out/prof_trace gpurun_out/summary_trace > /dev/null
python tools/pmc_summary.py gpurun_out/prof_fetch gpurun_out/summary_fetch > /dev/null
python tools/pmc_summary.py gpurun_out/prof_write gpurun_out/summary_write > /dev/null
cat gpurun_out/summary_trace.json | head -60
python -c "
import json
f=json.load(open('gpurun_out/summary_fetch.json'))['counters']; w=json.load(open('gpurun_out/summary_write.json'))['counters']
for k in sorted(set(f)|set(w)): print(k, f.get(k), w.get(k))
"
find gpurun_out/prof_trace -name '*stats*.csv' | head -3
# keep only the summaries + stats CSVs small enough to merge back
find gpurun_out/prof_fetch gpurun_out/prof_write -name '*.csv' -size +2M -delete
find gpurun_out/prof_trace -name '*kernel_trace.csv' -size +8M -delete
