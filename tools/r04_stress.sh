#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3; do
STEPS=400 timeout 300 python3 tools/dense_probe.py "" 2>&1 | grep -v amdgpu.ids | tail -2
DATA=clustered STEPS=1500 timeout 300 python3 tools/dense_probe.py "" 2>&1 | grep -v amdgpu.ids | tail -2
done
