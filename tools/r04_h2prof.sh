#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/ks_h2
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_h2 -o p -- python3 $GRAFT_REPO_ROOT/tools/h2_bench.py 1000000 768 clustered 64 > /tmp/ks_h2.log 2>&1)
f=$(find /tmp/ks_h2 -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 40 > gpurun_out/r04_h2_kernel_stats.txt
grep -v amdgpu /tmp/ks_h2.log | grep "build\|search" > gpurun_out/r04_h2_bench.log
head -14 gpurun_out/r04_h2_kernel_stats.txt | cut -c1-150; cat gpurun_out/r04_h2_bench.log
