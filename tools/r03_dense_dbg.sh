#!/bin/bash
cd $GRAFT_REPO_ROOT
for args in "--steps 10 --warmup 2 --recall-queries 0" "--steps 3 --warmup 1" "--steps 10 --warmup 2"; do for qb in 4 8; do
timeout 600 python3 bench.py --data gauss $args --hnsw-nvec 0 --gauss-steps 0 --build-from-host 0 --cpu-seconds 0 --opt screen16c_qb=$qb 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$args qb $qb: sweep', d['roofline'].get('avg_launch_ms'), 'ms; step', d['ms_per_step'], d['library_stats']['scan_launches'], d['library_stats']['rows_emitted'])"
done; done
