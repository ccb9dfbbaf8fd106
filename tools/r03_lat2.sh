#!/bin/bash
# parity of everything that selects probes, then the single-query kernel statistics
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_ivf.py tests/test_gpu_screen16.py tests/test_gpu_am.py tests/test_gpu_dist.py -x -q -m gpu > gpurun_out/lat2_tests.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/lat2_tests.log | head
bash tools/r03_lat_prof.sh 2>&1 | grep -E "k_ivf_scan|k_probe_select|k_merge_topk|k_ivf_topk|k_rows_scan|k_sum_candidates|copyBuffer|single-query|batch of"
timeout 300 python3 tools/latency.py 2>&1 | grep -v amdgpu.ids | tail -4
