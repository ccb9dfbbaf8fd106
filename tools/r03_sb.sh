#!/bin/bash
# tools/r03_sb.sh TAG: screen16 tests, small-batch latencies, single-query latency, headline step
tag=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_fullsize.py -x -q -m gpu 2>&1 | tail -3
NQS=1,8,16,32,64,128,256,512,1024 timeout 600 python3 tools/small_batch_probe.py > gpurun_out/${tag}_sb.log 2>&1
grep -E "nq=|screen_min" gpurun_out/${tag}_sb.log
timeout 300 python3 tools/latency.py > gpurun_out/${tag}_lat.log 2>&1; tail -4 gpurun_out/${tag}_lat.log
bash tools/r03_kprof.sh ${tag}_4096 2>&1 | grep -v "^#" | head -14
python3 -c "
import json
d=json.loads(open('gpurun_out/prof_${tag}_4096_bench_line.json').read().strip().splitlines()[-1]); print('C2 under the profiler', d['value'], d['ms_per_step'])"
