#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu -k dense_tile > gpurun_out/r04j_tests.log 2>&1
tail -3 gpurun_out/r04j_tests.log
timeout 900 python3 tools/dense_probe.py "screen16c_dense=0" "" "screen16c_sample=0" "screen16c_pfd=3" "screen16c_rot=1" "screen16_debug=6" "screen16_debug=2" "" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04j_dense.txt
