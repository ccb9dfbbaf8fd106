#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/ks_c5
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks_c5 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --nvec 10000000 --dim 1536 --rows f16 --strategy ip --batch 256 --lists 4096 --components 4096 --steps 40 --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 > /tmp/ks_c5.log 2>&1)
f=$(find /tmp/ks_c5 -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 70 > gpurun_out/r04_c5_kernel_stats.txt
grep -E " 4[0-9] +[0-9]" gpurun_out/r04_c5_kernel_stats.txt | cut -c1-132 | head -40
