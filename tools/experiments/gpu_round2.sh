#!/bin/bash
# One GPU-box visit of round 2: gpu tests, bench (headline + other configs), stage times, next rows, rocprofv3 kernel trace + PMC passes.
# Outputs under gpurun_out/r02_final/ (copied to profiles/r02_final by hand).
O=gpurun_out/r02_final
mkdir -p $O
R=$PWD
python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench.err; cut -c1-260 $O/bench_n1.json; tail -3 $O/bench.err
python tests/gpu_quick.py > $O/serial_stage_times_and_parity.log 2>&1; grep -E "PARITY|^B |single" $O/serial_stage_times_and_parity.log
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err; tail -2 $O/next_rows.err
for c in mono fisheye rgbd; do python bench.py --config $c --steps 60 --warmup 6 > $O/bench_$c.json 2>> $O/bench.err; done
python tools/soak_reference.py 4 300 > $O/soak_vs_reference.txt 2>&1; tail -2 $O/soak_vs_reference.txt
export TMPDIR=/tmp
PROF="--steps 12 --warmup 3 --no-cpu-baseline --no-h2d"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -o trace -- python $R/bench.py $PROF > $R/$O/prof_trace.log 2>&1)
PMC="--steps 4 --warmup 2 --pairs 64 --handles 3 --no-cpu-baseline --no-h2d"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o fetch -- python $R/bench.py $PMC > $R/$O/prof_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o write -- python $R/bench.py $PMC > $R/$O/prof_write.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/prof_sq -o sq -- python $R/bench.py $PMC > $R/$O/prof_sq.log 2>&1)
python tools/pmc_summary.py $O/prof_trace $O/kernel_trace_summary > /dev/null
python tools/pmc_summary.py $O/prof_fetch $O/pmc_fetch_size > /dev/null
python tools/pmc_summary.py $O/prof_write $O/pmc_write_size > /dev/null
python tools/pmc_summary.py $O/prof_sq $O/pmc_sq_counters > /dev/null
python tools/make_pmc_traffic.py $O/pmc_fetch_size.json $O/pmc_write_size.json $O/pmc_traffic.json $O/pmc_sq_counters.json $O/pmc_valu.json
find $O/prof_trace -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats.csv \;
head -12 $O/rocprofv3_kernel_stats.csv
# keep only the small summaries
rm -rf $O/prof_fetch $O/prof_write $O/prof_sq; find $O/prof_trace -name '*.csv' -size +4M -delete
