#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 tools/dense_probe.py "" "screen16c_rot=2" "screen16c_rot=1" "screen16_debug=6" "screen16_debug=6,screen16c_rot=2" "screen16_debug=3" "screen16_debug=3,screen16c_rot=2" "screen16c_rot=2,screen16c_pfd=2" "" "screen16c_rot=2" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04m_dense.txt
