#!/bin/bash
# Round 5: the matrix-core blur, second visit: A/B again (XCD-aware mapping), per-kernel times of both forms from a rocprofv3 kernel trace, traffic from PMC passes.
O=gpurun_out/r05_blur
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
AB="--steps 200 --warmup 20 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency"
for m in 1 0; do
  ORBX_BLUR_MODE=$m python bench.py $AB > $O/blur${m}_v2.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/blur${m}_v2.json')); print('ORBX_BLUR_MODE=$m', r['value'], r['ms_per_step'], r['parity_check']['identical'], 'blur alone', r['stage_ms_alone']['blur'], 'in flight', r['stage_ms_per_step']['blur'])"
done
PROF="--steps 12 --warmup 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-parity-check --min-seconds 0"
(cd /tmp && ORBX_BLUR_MODE=0 ORBX_BENCH_SERIAL=1 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_serial -o trace -- python $R/bench.py $PROF --handles 1 > $R/$O/prof_serial.log 2>&1)
find $O/prof_serial -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats_mfma_serial.csv \;
grep -E "k_blur|k_fast_cells|k_resize" $O/kernel_stats_mfma_serial.csv | cut -c1-60,150-260
PMC="--steps 4 --warmup 2 --pairs 64 --handles 1 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-parity-check --min-seconds 0"
for c in FETCH_SIZE WRITE_SIZE "SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR"; do
  n=$(echo $c | cut -d' ' -f1)
  (cd /tmp && ORBX_BLUR_MODE=0 rocprofv3 --pmc $c --output-format csv -d $R/$O/pmc_$n -o pmc -- python $R/bench.py $PMC > $R/$O/pmc_$n.log 2>&1)
  python tools/pmc_summary.py $O/pmc_$n $O/pmc_$n > /dev/null
  python -c "import json; r=json.load(open('$O/pmc_$n.json'))['counters']; print('$n', {k: v for k, v in r.items() if 'blur' in k})"
done
rm -rf $O/prof_serial $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_SQ_INSTS_VALU
