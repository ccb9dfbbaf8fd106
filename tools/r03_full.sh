#!/bin/bash
# tools/r03_full.sh TAG: whole GPU suite, then the default bench line (all legs)
tag=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
tail -3 gpurun_out/${tag}_tests.log
timeout 1500 python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.log </dev/null
tail -2 gpurun_out/${tag}_bench.log
python3 - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_bench_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('C2', d['value'], d['ms_per_step'], r.get('bound'), r.get('frac'), r.get('avg_launch_ms'), 'hbm', r.get('hbm',{}).get('frac'), 'step_frac', r.get('hbm',{}).get('step_frac'))
print('cpu', d.get('cpu_baseline'))
b=d.get('build') or {}
print('build', {k:b.get(k) for k in ('vectors_per_s','seconds','prepare_seconds','searchable_vectors_per_s','cpu_baseline')})
g=d.get('iid_gauss') or {}
print('iid', {k:g.get(k) for k in ('queries_per_s','ms_per_step','recall_at_10','cpu_baseline','error')}, (g.get('roofline') or {}).get('frac'), (g.get('roofline') or {}).get('bound'))
g=d.get('balanced_index') or {}
print('bal', {k:g.get(k) for k in ('queries_per_s','ms_per_step','recall_at_10','error')})
h=d.get('hnsw') or {}
print('hnsw', {k:h.get(k) for k in ('queries_per_s','recall_at_10','build_vectors_per_s','roofline','error')})
print('intended', json.dumps(h.get('intended'))[:1500])
PY
