#!/bin/bash
# the service with plain-C backends (examples/service_clients.c) on the headline table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r04_service_bench_c.txt
timeout 600 python3 -m pytest tests/test_service.py -x -q 2>&1 | tail -2
for cfg in "1 1 2000 4096 100 256" "16 16 4000 4096 20 512" "32 32 4000 4096 20 2048" "32 32 4000 4096 20 8192" "32 32 4000 4096 5 2048" "32 64 6000 4096 20 4096" "64 32 4000 4096 20 4096" "64 64 4000 4096 20 8192"; do
set -- $cfg
timeout 900 python3 tools/service_bench.py --clients c --backends $1 --inflight $2 --queries $3 --max-batch $4 --linger-us $5 --n 1000000 --dim 768 --nlists 1024 --nprobe 32 --check 8 --nslots $6 --data c2 2>&1 | grep -v amdgpu | tail -1 | tee -a gpurun_out/r04_service_bench_c.txt | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$cfg', d['aggregate_queries_per_s'], d['owner'], d['avg_batch'], d['mismatches'])"
done
