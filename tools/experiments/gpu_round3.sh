#!/bin/bash
# One GPU-box visit of round 3: gpu tests, bench lines (headline, 20-step driver form, natural workload, all-gather at N = 1, import-copy variant, other
# configs), stage times, next rows, rocprofv3 kernel trace + PMC passes.  Outputs under gpurun_out/r03_final/ (copied to profiles/r03_final by hand).
O=gpurun_out/r03_final
mkdir -p $O
R=$PWD
python -m pytest tests -x -q -m gpu -s > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log; grep "ORBvoc-scale" $O/pytest_gpu.log
python bench.py > $O/bench_n1.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1.json')); print('headline', r['value'], r['ms_per_step'], r['repeats'], r['block_values']['min'], r['block_values']['max'], r['roofline']['frac'], r['roofline']['alone_launch_ms'], r.get('h2d_inclusive',{}).get('value'), r['cpu_baseline']['value'])"; tail -3 $O/bench.err
python bench.py --steps 20 --warmup 5 --no-h2d --no-cpu-baseline > $O/bench_20steps.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_20steps.json')); print('20 steps:', r['value'], r['repeats'], r['block_values']['min'], r['block_values']['max'])"
python bench.py --import-copy --steps 100 --warmup 10 --no-h2d --no-cpu-baseline > $O/bench_import_copy.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_import_copy.json')); print('import-copy:', r['value'])"
python bench.py --workload natural --steps 100 --warmup 10 --no-cpu-baseline --no-h2d > $O/bench_natural.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_natural.json')); print('natural', r['value'], r['config']['fast_corner_density_t7'], r['roofline']['frac'], r['stage_ms_alone'])"
python bench.py --allgather --steps 100 --warmup 10 --no-cpu-baseline 2>> $O/bench.err | grep "^{" > $O/bench_allgather_n1.json; python -c "import json; r=json.load(open('$O/bench_allgather_n1.json')); print('allgather', r['value'], r['allgather'])"
for c in mono fisheye rgbd; do python bench.py --config $c --steps 60 --warmup 6 --no-cpu-baseline > $O/bench_$c.json 2>> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_$c.json')); print('$c', r['value'], r['ms_per_step'])"; done
python tests/gpu_quick.py > $O/serial_stage_times_and_parity.log 2>&1; grep -E "PARITY|^B |single" $O/serial_stage_times_and_parity.log
python tools/bench_next_rows.py > $O/next_rows.json 2> $O/next_rows.err; tail -2 $O/next_rows.err
python tools/soak_reference.py 4 300 > $O/soak_vs_reference.txt 2>&1; tail -2 $O/soak_vs_reference.txt
python tools/soak_round3.py 12 > $O/soak_round3.txt 2>&1; tail -5 $O/soak_round3.txt
export TMPDIR=/tmp
PROF="--steps 12 --warmup 3 --no-cpu-baseline --no-h2d --min-seconds 0"
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace -o trace -- python $R/bench.py $PROF > $R/$O/prof_trace.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_trace_rgbd -o trace -- python $R/bench.py --config rgbd $PROF > $R/$O/prof_trace_rgbd.log 2>&1)
PMC="--steps 4 --warmup 2 --pairs 64 --handles 3 --no-cpu-baseline --no-h2d --min-seconds 0"
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
# keep only the small summaries
rm -rf $O/prof_fetch $O/prof_write $O/prof_sq; find $O/prof_trace $O/prof_trace_rgbd -name '*.csv' -size +4M -delete
# second half of round 3: per-call latency of the drop-in facade, kernel trace of the single-pair loop, launch / wait latencies, pyramid launch forms
bash tools/gpu_facade_latency.sh > /dev/null 2>&1; cp gpurun_out/facade_latency.txt $O/facade_latency.txt 2>/dev/null; cat $O/facade_latency.txt
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_single -o trace -- python $R/tools/single_pair_loop.py 300 > $R/$O/prof_single.log 2>&1); tail -1 $O/prof_single.log
find $O/prof_single -name '*kernel_stats.csv' -exec cp {} $O/rocprofv3_kernel_stats_single_pair.csv \; ; find $O/prof_single -name '*.csv' -size +2M -delete
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -w tools/sync_latency_probe.hip -o /tmp/sync_latency_probe && /tmp/sync_latency_probe > $O/sync_latency_probe.txt; cat $O/sync_latency_probe.txt
python tools/time_pyramid_modes.py 2 8 16 > $O/pyramid_one_launch_vs_per_level.txt 2>&1; cat $O/pyramid_one_launch_vs_per_level.txt
