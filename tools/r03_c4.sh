#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_fullsize.py tests/test_gpu_dist.py tests/test_gpu_hnsw.py tests/test_gpu_build.py tests/test_gpu_am.py tests/test_gpu_sql.py tests/test_service.py -x -q -m gpu > gpurun_out/r03m_tests.log 2>&1
grep -n "passed\|failed" gpurun_out/r03m_tests.log | tail -2
NVEC=10000000 LISTS=4096 timeout 900 python3 tools/shard_step_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_shard_step_probe_c4.log
timeout 900 python3 tools/shard_step_probe.py 2>&1 | grep -v amdgpu | tee gpurun_out/r03_shard_step_probe_c2.log
