#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_h2
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_h2 -o p -- python3 $GRAFT_REPO_ROOT/tools/h2_bench.py 200000 768 clustered 64 > /tmp/prof_h2.log 2>&1
grep -v amdgpu /tmp/prof_h2.log | tail -3
f=$(find /tmp/prof_h2 -name "*.db" | head -1)
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f 12 | cut -c1-60,73-130
