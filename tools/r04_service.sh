#!/bin/bash
# the device-owner service on the headline table (--data c2) and on the random test index of the earlier rounds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
: > gpurun_out/r04_service_bench.txt
for data in c2 random; do
for cfg in "1 1 400" "16 1 400" "16 16 2000" "32 32 2000" "32 64 4000"; do
set -- $cfg
timeout 900 python3 tools/service_bench.py --backends $1 --inflight $2 --queries $3 --n 1000000 --dim 768 --nlists 1024 --nprobe 32 --check 16 --nslots 4096 --data $data 2>&1 | grep -v amdgpu | tail -1 | tee -a gpurun_out/r04_service_bench.txt | cut -c1-330
done
done
