#!/bin/bash
# tests + default bench of the current binary:  tools/r03_run.sh TAG [extra bench args]
tag=$1; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python3 -m pytest tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
tail -6 gpurun_out/${tag}_tests.log
timeout 900 python3 bench.py --hnsw-nvec 0 "$@" > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.log </dev/null
tail -3 gpurun_out/${tag}_bench.log
python3 - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_bench_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('C2', d['value'], d['ms_per_step'], r.get('kernel'), r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'))
print('stats', d['library_stats'])
b=d.get('build') or {}
print('build', {k:b.get(k) for k in ('vectors_per_s','seconds','prepare_seconds','searchable_vectors_per_s')})
for leg in ('iid_gauss','balanced_index'):
    g=d.get(leg)
    if g: print(leg, {k:g.get(k) for k in ('queries_per_s','ms_per_step','recall_at_10','rows_rescored_per_query','screen16','oracle_parity','error')})
print('cpu', (d.get('cpu_baseline') or {}).get('gpu_parity_on_sample'))
PY
