#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 tools/dense_probe.py "" "screen16c_pf=16" "screen16c_pf=32" "screen16c_pf=48" "" "screen16c_pf=16" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04d_nt.txt
