#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/ks_c2
(cd /tmp && DATA=clustered STEPS=20 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks_c2 -o p -- python3 $GRAFT_REPO_ROOT/tools/dense_probe.py "" > /tmp/ks_c2.log 2>&1)
f=$(find /tmp/ks_c2 -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 60 > gpurun_out/r04_c2_kernel_stats.txt
grep -v amdgpu.ids /tmp/ks_c2.log | tail -1
grep -E " 2[0-9] | 4[0-9] " gpurun_out/r04_c2_kernel_stats.txt | cut -c1-130 | head -40
