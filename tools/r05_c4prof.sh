#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rm -rf /tmp/ks6; (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/ks6 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --nvec 10000000 --lists 4096 --steps 5 --warmup 2 --inflight 1 --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --c4-nvec 0 --sigma-sweep 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 "$@" > /tmp/ks6.log 2>&1); tail -2 /tmp/ks6.log | cut -c1-400
f=$(find /tmp/ks6 -name "*.db" | head -1); [ -n "$f" ] && python3 tools/rocpd_summary.py $f 60 > gpurun_out/r05_kstats_c4.txt; grep -E "k_s16|k_cent|k_sub|k_pair|k_probe|fillBuffer|copyBuffer" gpurun_out/r05_kstats_c4.txt | cut -c1-75,76-150 | head -36
