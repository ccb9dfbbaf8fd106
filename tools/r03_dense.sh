#!/bin/bash
# tools/r03_dense.sh TAG: the i.i.d. table with the 128 x 128 and the 256 x 256 tile of the centred sweep (one box), CPU parity sample each
tag=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for qb in ${QBS:-4 8}; do
timeout 900 python3 bench.py --data gauss --steps 10 --warmup 2 --hnsw-nvec 0 --gauss-steps 0 --build-from-host 0 --cpu-seconds ${CPUS:-4} --opt screen16c_qb=$qb > gpurun_out/${tag}_qb${qb}.json 2> gpurun_out/${tag}_qb${qb}.log </dev/null
python3 - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_qb${qb}.json').read().strip().splitlines()[-1])
r=d['roofline']
print('qb $qb:', d['value'], 'q/s', d['ms_per_step'], 'ms/step; sweep', r.get('avg_launch_ms'), 'ms, mfma frac', r['mfma']['frac'], 'recall', d['recall_at_10'], 'cpu parity', (d.get('cpu_baseline') or {}).get('gpu_parity_on_sample'), 'fallbacks', d['library_stats']['screen16_fallbacks'])
PY
done
