#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/prof_pass.sh r03c_clustered 2>&1 | cut -c1-118 | head -34
bash tools/prof_pass.sh r03c_gauss --data gauss 2>&1 | cut -c1-118 | head -24
cd $GRAFT_REPO_ROOT
for v in "screen16c_qb=4" "screen16c_qb=1 --opt screen16c_nbuf=2" "screen16_centered=0"; do
  echo "== clustered $v"
  timeout 600 python3 bench.py --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --recall-queries 0 --steps 30 --opt $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
done
for v in "screen16c_nbuf=3" "screen16c_qb=1" "screen16_centered=0"; do
  echo "== gauss $v"
  timeout 600 python3 bench.py --data gauss --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --recall-queries 0 --steps 10 --opt $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'))"
done
