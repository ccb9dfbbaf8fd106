#!/bin/bash
# parity of the screened scan, the quick bench (L2 and cosine), per-kernel statistics of the headline step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_dist.py tests/test_gpu_ivf.py -x -q -m gpu > gpurun_out/qp_tests.log 2>&1; grep -E "passed|failed|Error|assert" gpurun_out/qp_tests.log | head
B="--hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --steps 20 --warmup 5"
for s in l2 cosine ip; do
timeout 600 python3 bench.py $B --strategy $s 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$s', d['value'], d['ms_per_step'])"
done
bash tools/r03_kprof.sh qp 2>&1 | head -24
