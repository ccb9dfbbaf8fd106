#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python3 bench.py --nvec 200000 --lists 256 --components 256 --steps 2 --warmup 1 --cpu-seconds 0 --hnsw-nvec 0 --gauss-steps 0 --recall-queries 16 --opt debug_s16=1 --opt screen16_sub_min=256 > gpurun_out/r03dbg.json 2> gpurun_out/r03dbg.log </dev/null
grep "s16" gpurun_out/r03dbg.log | head -12
tail -c 400 gpurun_out/r03dbg.json
timeout 900 python3 bench.py --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --opt screen16_sub_min=256 > gpurun_out/r03b_bench_line.json 2> gpurun_out/r03b_bench.log </dev/null
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r03b_bench_line.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline'].get('kernel'), d['roofline'].get('avg_launch_ms'), d['library_stats'])
PY
