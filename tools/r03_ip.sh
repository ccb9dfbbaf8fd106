#!/bin/bash
# inner product on the C2 table (float4 rows), with and without sublists; then a fuzz campaign
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for o in "screen16_sublists=1" "screen16_sublists=0"; do
timeout 600 python3 bench.py --strategy ip --hnsw-nvec 0 --gauss-steps 0 --build-from-host 0 --cpu-seconds ${CPUS:-4} --opt $o "$@" 2>gpurun_out/r03_ip.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; st=d['library_stats']
print('ip $o:', d['value'], 'q/s', d['ms_per_step'], 'ms; recall', d['recall_at_10'], 'cpu parity', (d.get('cpu_baseline') or {}).get('gpu_parity_on_sample'), 'swept frac', round(st['rows_swept']/max(1,st['rows_scored']),4), 'rescored/q', r.get('rows_rescored_per_query'), 'fallbacks', st['screen16_fallbacks'])"
done
timeout 300 python3 tools/fuzz_scan.py ${FUZZ:-150} 21 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
