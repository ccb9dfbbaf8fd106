#!/bin/bash
# tools/r03_host.sh TAG: the build tests, then the bench with the from-host build leg
tag=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_build.py tests/test_golden.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
tail -3 gpurun_out/${tag}_tests.log
timeout 600 python3 bench.py --opt debug_build=1 --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.log </dev/null
grep -E "build:" gpurun_out/${tag}_bench.log | tail -24
python3 - <<PY
import json
d=json.loads(open('gpurun_out/${tag}_bench_line.json').read().strip().splitlines()[-1])
print('C2', d['value'], d['ms_per_step'])
b=d['build']; print({k:b[k] for k in ('vectors_per_s','seconds','prepare_seconds','searchable_vectors_per_s','from_host')})
PY
