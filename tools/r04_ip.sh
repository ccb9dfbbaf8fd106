#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_ivf.py -x -q -m gpu > gpurun_out/r04_ip_tests.log 2>&1
tail -4 gpurun_out/r04_ip_tests.log
for v in 1 0; do
timeout 600 python3 bench.py --strategy ip --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --opt screen16_ip_centered=$v 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('ip centered=$v', d['value'], d['ms_per_step'], r.get('kernel','')[:30], r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'), 'emitted', r.get('rows_emitted_per_query'), 'fallbacks', d['library_stats']['screen16_fallbacks'])"
done
