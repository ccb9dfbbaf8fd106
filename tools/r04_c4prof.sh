#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/ks_c4
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks_c4 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --nvec 10000000 --lists 4096 --components 4096 --steps 20 --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 > /tmp/ks_c4.log 2>&1)
f=$(find /tmp/ks_c4 -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 80 > gpurun_out/r04_c4_kernel_stats.txt
grep -E " +2[0-9] +[0-9.]+ +[0-9.]+ " gpurun_out/r04_c4_kernel_stats.txt | cut -c1-140 | head -30
tail -1 /tmp/ks_c4.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), d['roofline'].get('frac'))"
