#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 900 python3 tools/dense_probe.py "" "screen16c_pfd=17" "screen16c_pfd=18" "screen16c_pfd=19" "screen16c_pfd=20" "screen16c_pfd=22" 2>&1 | grep -v amdgpu
