#!/bin/bash
# round 5: after the Fuse gate fix (k_search.hip) - the search / matcher tests on the GPU and the three PMC passes that stamp profiles/pmc_traffic.json with the new device sources
O=gpurun_out/r05_fusefix
mkdir -p $O
R=$PWD
export TMPDIR=/tmp
python -m pytest tests -q -m gpu -k "search or matcher or resident or headline_shape_stereo" > $O/pytest.log 2>&1; grep -E "passed|failed|^E " $O/pytest.log | tail -3
PMC="--steps 4 --warmup 2 --pairs 64 --handles 3 --no-cpu-baseline --no-h2d --no-other-configs --no-latency --no-live-traffic --min-seconds 0"
(cd /tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/$O/prof_fetch -o fetch -- python $R/bench.py $PMC > $R/$O/prof_fetch.log 2>&1)
(cd /tmp && rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/$O/prof_write -o write -- python $R/bench.py $PMC > $R/$O/prof_write.log 2>&1)
(cd /tmp && rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT --output-format csv -d $R/$O/prof_sq -o sq -- python $R/bench.py $PMC > $R/$O/prof_sq.log 2>&1)
python tools/pmc_summary.py $O/prof_fetch $O/pmc_fetch_size > /dev/null
python tools/pmc_summary.py $O/prof_write $O/pmc_write_size > /dev/null
python tools/pmc_summary.py $O/prof_sq $O/pmc_sq_counters > /dev/null
python tools/make_pmc_traffic.py $O/pmc_fetch_size.json $O/pmc_write_size.json $O/pmc_traffic.json $O/pmc_sq_counters.json $O/pmc_valu.json
rm -rf $O/prof_fetch $O/prof_write $O/prof_sq
python -c "import json; t=json.load(open('$O/pmc_traffic.json')); print({k: t[k] for k in list(t)[:6]})" | cut -c1-400
