#!/bin/bash
# the whole GPU suite, then the single-query kernel statistics and the quick L2 line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/lat3_tests.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/lat3_tests.log | head
bash tools/r03_lat_prof.sh 2>&1 | grep -E "k_ivf_scan<|k_probe_select|k_merge_topk|k_ivf_topk|k_rows_scan|k_sum_candidates|copyBuffer|single-query|batch of|k_s16_finalize"
timeout 300 python3 tools/latency.py 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python3 bench.py --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('l2', d['value'], d['ms_per_step'])"
