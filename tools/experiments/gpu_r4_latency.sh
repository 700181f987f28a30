#!/bin/bash
# round 4, latency of small batches: the small-batch launch forms (blur + FAST in one launch, event records on demand) against the large-batch forms
# on one pair / one image per call, the parity tests of the forms, and a kernel trace of the single-pair chain
O=gpurun_out/r04_latency
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
timeout 600 python -m pytest tests/test_emu_parity.py tests/test_gpu_parity.py tests/test_frame_reference.py -m gpu -x -q 2>&1 | tail -4 | tee $O/pytest.log
timeout 300 python tools/single_pair_loop.py 400 ab 2>&1 | tee $O/single_pair_ab.txt
for f in 1 0; do
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_forms$f -o trace -- python tools/single_pair_loop.py 150 $f > $O/prof_forms$f.log 2>&1
  cp $O/prof_forms$f/trace_kernel_stats.csv $O/kernel_stats_forms$f.csv 2>/dev/null
done
python - <<'P' | tee $O/timeline.txt
import csv, glob
for f in (1, 0):
    p = glob.glob('gpurun_out/r04_latency/prof_forms%d/**/trace_kernel_trace.csv' % f, recursive=True) + glob.glob('gpurun_out/r04_latency/prof_forms%d/trace_kernel_trace.csv' % f)
    if not p: print("no trace", f); continue
    rows = sorted(csv.DictReader(open(p[0])), key=lambda r: int(r['Start_Timestamp']))
    i0 = len(rows) // 2
    while 'pyramid' not in rows[i0]['Kernel_Name']: i0 += 1
    t0 = int(rows[i0]['Start_Timestamp'])
    print("forms", f)
    for r in rows[i0:i0 + 9]:
        s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
        print("  %-26s start %7.1f end %7.1f dur %6.1f" % (r['Kernel_Name'].split('(')[0][6:], s / 1e3, e / 1e3, (e - s) / 1e3))
P
