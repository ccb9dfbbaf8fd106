#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for cfg in "32 32 4000 4096 20 2048" "16 64 8000 4096 20 2048" "8 128 16000 4096 20 2048" "32 64 6000 4096 20 4096" "16 128 12000 4096 20 4096"; do
set -- $cfg
timeout 900 python3 tools/service_bench.py --clients c --backends $1 --inflight $2 --queries $3 --max-batch $4 --linger-us $5 --n 1000000 --dim 768 --nlists 1024 --nprobe 32 --check 8 --nslots $6 --data c2 2>&1 | grep -v amdgpu | tail -1 | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['aggregate_queries_per_s'], d['owner'], d['avg_batch'], d['mismatches'])"
done
