#!/bin/bash
# round 6, profile part of the evidence run: rocprofv3 kernel trace + PMC passes of the bench command (no latency loop, no CPU baseline)
O=gpurun_out/r06_final
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
PROF="--steps 12 --warmup 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --min-seconds 0"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -o trace -- python $R/bench.py $PROF > $R/$O/prof_trace.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_rgbd -o trace -- python $R/bench.py --config rgbd $PROF > $R/$O/prof_trace_rgbd.log 2>&1)
PMC="--steps 4 --warmup 2 --pairs 64 --handles 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --min-seconds 0"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o fetch -- python $R/bench.py $PMC > $R/$O/prof_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o write -- python $R/bench.py $PMC > $R/$O/prof_write.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/prof_sq -o sq -- python $R/bench.py $PMC > $R/$O/prof_sq.log 2>&1)
python tools/pmc_summary.py $O/prof_trace $O/kernel_trace_summary > /dev/null
python tools/pmc_summary.py $O/prof_fetch $O/pmc_fetch_size > /dev/null
python tools/pmc_summary.py $O/prof_write $O/pmc_write_size > /dev/null
python tools/pmc_summary.py $O/prof_sq $O/pmc_sq_counters > /dev/null
python tools/make_pmc_traffic.py $O/pmc_fetch_size.json $O/pmc_write_size.json $O/pmc_traffic.json $O/pmc_sq_counters.json $O/pmc_valu.json
find $O/prof_trace -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats.csv \;
find $O/prof_trace_rgbd -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats_rgbd.csv \;
head -12 $O/rocprofv3_kernel_stats.csv | cut -c1-200
rm -rf $O/prof_fetch $O/prof_write $O/prof_sq; find $O/prof_trace $O/prof_trace_rgbd -name '*.csv' -size +4M -delete
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single -o trace -- python $R/tools/single_pair_loop.py 300 > $R/$O/prof_single.log 2>&1); tail -1 $O/prof_single.log
find $O/prof_single -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats_single_pair.csv \; ; find $O/prof_single -name '*.csv' -size +2M -delete
