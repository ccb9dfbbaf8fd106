#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in 1 0; do
timeout 1200 python3 bench.py --nvec 10000000 --dim 1536 --rows f16 --strategy ip --batch 256 --lists 4096 --components 4096 --steps 20 --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --opt screen16_ip_centered=$v 2> gpurun_out/r04_c5_$v.log | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('C5 10M x 1536 fp16 ip, B=256 centered=$v', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'), 'emitted', r.get('rows_emitted_per_query'), 'fallbacks', d['library_stats']['screen16_fallbacks'], 'build', d['build_vectors_per_s'])"
grep -v amdgpu gpurun_out/r04_c5_$v.log | tail -3
done
