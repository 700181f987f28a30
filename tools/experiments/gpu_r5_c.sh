#!/bin/bash
# Round 5, third GPU-box visit: ORBX_SCHED A/B (stream priorities / head of the chain on a low-priority stream), then the full default bench line
# with the reworked baselines (cpu all-core processes, PCIe probe, other_configs incl. natural imagery and fisheye).
O=gpurun_out/r05_c
mkdir -p $O
AB="--steps 200 --warmup 20 --min-seconds 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency"
for rep in 1 2; do
for m in 0 1 2; do
  ORBX_SCHED=$m python bench.py $AB > $O/sched${m}_$rep.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/sched${m}_$rep.json')); print('ORBX_SCHED=$m rep $rep', r['value'], r['ms_per_step'], r['parity_check']['identical'], {k: round(v, 3) for k, v in r['stage_ms_per_step'].items()})"
done
done
for m in 0 2; do
  ORBX_SCHED=$m python bench.py $AB --workload natural > $O/sched${m}_natural.json 2>> $O/ab.err
  python -c "import json; r=json.load(open('$O/sched${m}_natural.json')); print('natural ORBX_SCHED=$m', r['value'], r['ms_per_step'], r['parity_check']['identical'])"
done
timeout 900 python bench.py > $O/bench_n1.json 2> $O/bench.err; echo "bench rc $?"
python -c "
import json; r=json.load(open('$O/bench_n1.json'))
print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r['parity_check']['identical'])
print('cpu', {k: (r['cpu_baseline'][k]['value'], r['cpu_baseline'][k].get('scaling_efficiency_vs_two_cores'), r['cpu_baseline'][k].get('late_starters')) for k in ('one_core','two_cores','all_cores')})
print('h2d', r['h2d_inclusive'])
print('other', {k: (v.get('value'), (v.get('parity_check') or {}).get('identical'), v.get('timed_seconds')) for k, v in r['other_configs'].items()})
"; tail -3 $O/bench.err
