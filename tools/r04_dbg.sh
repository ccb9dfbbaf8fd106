#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 tools/dense_probe.py "" "screen16_debug=3" "screen16_debug=4" "screen16_debug=2" "screen16_debug=1" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04c_dbg.txt
rocprofv3 -L > gpurun_out/r04_counters.txt 2>&1
wc -l gpurun_out/r04_counters.txt
