#!/bin/bash
# the whole GPU suite, the quick lines of the three metrics, kernel statistics of the headline step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/full_tests.log 2>&1; grep -E "passed|failed|Error|^E " gpurun_out/full_tests.log | head
B="--hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --steps 20 --warmup 5"
for s in l2 cosine ip; do
timeout 600 python3 bench.py $B --strategy $s 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', d['value'], d['ms_per_step'])"
done
bash tools/r03_kprof.sh qp 2>&1 | head -16
