#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_ivf.py -x -q -m gpu > gpurun_out/r04_ip_tests.log 2>&1
tail -3 gpurun_out/r04_ip_tests.log
timeout 600 python3 tools/c5_check.py 1000000 1024 3 2>&1 | grep -v amdgpu | tail -6
for v in 1 0; do
for b in 256 4096; do
timeout 600 python3 bench.py --dim 1536 --rows f16 --strategy ip --batch $b --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --opt screen16_ip_centered=$v 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('C5 shape 1M x 1536 fp16 ip, B=$b centered=$v', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'), 'emitted', r.get('rows_emitted_per_query'), 'fallbacks', d['library_stats']['screen16_fallbacks'])"
done
done
