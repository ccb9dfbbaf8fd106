#!/bin/bash
# round 4: the dense sweep's L2 prefetch at several distances (i.i.d. table), same results required
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 tools/dense_probe.py "" "screen16c_pf=2" "screen16c_pf=3" "screen16c_pf=4" "screen16c_pf=6" "screen16_debug=1" "screen16_debug=2" "" "screen16c_pf=3" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04b_pf.txt
