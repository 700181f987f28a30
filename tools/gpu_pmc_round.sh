#!/bin/bash
# PMC pass: SQ instruction / stall counters per kernel (own run, no trace domains besides the implicit kernel dispatch)
mkdir -p gpurun_out; R=$PWD; export TMPDIR=/tmp
rm -rf gpurun_out/prof_sq
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/prof_sq -o sq -- python $R/bench.py --steps 3 --warmup 1 --pairs 64 --no-cpu-baseline > $R/gpurun_out/prof_sq.log 2>&1)
tail -2 gpurun_out/prof_sq.log | cut -c1-200
python tools/pmc_summary.py gpurun_out/prof_sq gpurun_out/summary_sq > /dev/null
python - <<'PY'
import json
c=json.load(open('gpurun_out/summary_sq.json'))['counters']
for k,v in sorted(c.items()):
    if not k.startswith('k_'): continue
    g=lambda n: v.get(n,{}).get('avg',0)
    w=g('SQ_WAVES') or 1
    print('%-16s waves %9.0f  valu/wave %7.0f salu/wave %6.0f lds/wave %6.0f  wave_cycles/wave %8.0f wait_inst_any %5.2f active_valu %5.2f bankconf/ldsinst %.3f' % (
        k, w, g('SQ_INSTS_VALU')/w, g('SQ_INSTS_SALU')/w, g('SQ_INSTS_LDS')/w, g('SQ_WAVE_CYCLES')/w*4, g('SQ_WAIT_INST_ANY')/max(g('SQ_WAVE_CYCLES'),1), g('SQ_ACTIVE_INST_VALU')/max(g('SQ_WAVE_CYCLES'),1), g('SQ_LDS_BANK_CONFLICT')/max(g('SQ_INSTS_LDS'),1)))
PY
find gpurun_out/prof_sq -name '*.csv' -size +2M -delete
