#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { timeout 1200 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], r.get('avg_launch_ms'), r.get('frac'), 'emitted', r.get('rows_emitted_per_query'), 'rescored', r.get('rows_rescored_per_query'), 'fallbacks', d['library_stats']['screen16_fallbacks'])" "$*"; }
run --steps 30
run --steps 30 --opt screen16c_seeds=16
run --steps 30 --opt screen16c_seeds=24
run --steps 30 --opt screen16c_seeds=48
