#!/bin/bash
# the driver's command, then the service, the small batches and the single query on the same binary
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 bench.py > gpurun_out/r04_bench_line.json 2> gpurun_out/r04_bench.log </dev/null
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r04_bench_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('C2', d['value'], d['ms_per_step'], 'sweep', r.get('avg_launch_ms'), 'frac', r['frac'], 'step_frac', r['hbm']['step_frac'], 'recall', d['recall_at_10'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline'].get('gpu_parity_on_sample'))
g=d['iid_gauss']
print('iid', d['value_iid'], g.get('ms_per_step'), g['roofline']['frac'], g['roofline'].get('avg_launch_ms'), g['recall_at_10'], g.get('oracle_parity'), g['roofline']['kernel'][:20])
h=d['hnsw']
print('hnsw', h.get('queries_per_s'), h.get('recall_at_10'), h.get('mode'), 'ref', h['ref_compat']['queries_per_s'] if 'ref_compat' in h else h)
print('build', d['build'])
PY
timeout 600 python3 tools/service_bench.py > gpurun_out/r04_service_bench.txt 2>&1; tail -12 gpurun_out/r04_service_bench.txt
timeout 300 python3 tools/small_batch_probe.py 2>&1 | grep -v amdgpu > gpurun_out/r04_small_batch.txt; cat gpurun_out/r04_small_batch.txt
timeout 300 python3 tools/latency.py --n 1000 2>&1 | grep -v amdgpu > gpurun_out/r04_latency.txt; cat gpurun_out/r04_latency.txt
