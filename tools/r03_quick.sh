#!/bin/bash
# tools/r03_quick.sh TAG "pytest targets" [bench args]: a few test files, then the clustered C2 step only
tag=$1; tests=$2; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest $tests -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
tail -12 gpurun_out/${tag}_tests.log
timeout 600 python3 bench.py --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 "$@" > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.log </dev/null
tail -3 gpurun_out/${tag}_bench.log
python3 - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_bench_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('C2', d['value'], d['ms_per_step'], r.get('kernel'), r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'), 'recall', d['recall_at_10'])
print('stats', d['library_stats'])
PY
