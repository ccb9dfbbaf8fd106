#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu -k "dense" 2>&1 | tail -2
timeout 900 python3 tools/dense_probe.py "" "screen16c_pfd=34" "screen16c_pfd=35" "screen16c_pfd=36" "screen16c_pfd=38" "screen16c_pfd=40" 2>&1 | grep -v amdgpu
