#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu -k "list_level_pruning" 2>&1 | grep -v "^$" | tail -40
rm -rf /tmp/ks5; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks5 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 > /tmp/ks5.log 2>&1)
f=$(find /tmp/ks5 -name "*.db" | head -1); [ -n "$f" ] && python3 tools/rocpd_summary.py $f 70 > gpurun_out/r05_kstats_c2.txt; grep -E "k_s16|k_cent|k_sub|k_pair|k_probe|fillBuffer|copyBuffer" gpurun_out/r05_kstats_c2.txt | cut -c1-75,76-140 | head -40
