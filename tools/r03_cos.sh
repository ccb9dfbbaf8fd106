#!/bin/bash
# cosine on the C2 table: the matrix-core sweep over normalised planes against the round-1 fp32 screen; fuzz; kernel profile
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for o in "screen16_cosine=1" "screen16_cosine=0"; do
timeout 600 python3 bench.py --strategy cosine --hnsw-nvec 0 --gauss-steps 0 --build-from-host 0 --cpu-seconds ${CPUS:-4} --opt $o "$@" 2>gpurun_out/r03_cos.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; st=d['library_stats']
print('cosine $o:', d['value'], 'q/s', d['ms_per_step'], 'ms; cpu parity', (d.get('cpu_baseline') or {}).get('gpu_parity_on_sample'), 'swept frac', round(st['rows_swept']/max(1,st['rows_scored']),4), 'rescored/q', r.get('rows_rescored_per_query'), 'fallbacks', st['screen16_fallbacks'], 'batches', st['screen16_batches'])"
done
timeout 400 python3 tools/fuzz_scan.py ${FUZZ:-200} 41 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-400
