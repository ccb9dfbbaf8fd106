#!/bin/bash
# tools/r03_var.sh "bench args common" "opt set 1" "opt set 2" ...: the C2 step under several option sets
cd $GRAFT_REPO_ROOT
common=$1; shift
for v in "$@"; do
  echo "== $v"
  timeout 600 python3 bench.py --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --recall-queries 0 --steps 30 $common $v 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); st=d['library_stats']; print(d['value'], d['ms_per_step'], d['roofline'].get('avg_launch_ms'), 'emitted/q', st['rows_emitted']/st['queries'], 'rescored/q', st['rows_rescored']/st['queries'], 'fallbacks', st['screen16_fallbacks'])"
done
