#!/bin/bash
# Round 6, last visit: the final tree's GPU suite, the node-pool fuzz soak on the device, host waits after PyTorch initialised the device, the default line.
O=gpurun_out/r06_last
mkdir -p $O
python -m pytest tests -x -q -m gpu > $O/pytest_gpu_final_tree.log 2>&1; grep -E "passed|failed" $O/pytest_gpu_final_tree.log | tail -2
python tools/soak_fuzz.py hip 21000 21399 pool > $O/soak_fuzz_node_pool_gpu.txt 2>&1; tail -1 $O/soak_fuzz_node_pool_gpu.txt
python tools/experiments/hostwait_after_torch.py > $O/hostwait_after_torch.txt 2>&1; tail -2 $O/hostwait_after_torch.txt
python bench.py > $O/bench_n1_final_tree.json 2> $O/bench.err; python -c "import json; r=json.load(open('$O/bench_n1_final_tree.json')); print('headline', r['value'], r['ms_per_step'], r['roofline']['frac'], r['roofline']['traffic_source']['stale'], r['roofline']['step_valu_issue'], r['parity_check']['identical'], r.get('value_host_fed'), {k: (v.get('value'), (v.get('parity_check') or {}).get('identical')) for k, v in r['other_configs'].items()})"; tail -2 $O/bench.err
