#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_screen16w.py tests/test_gpu_ivf.py -x -q -m gpu 2>&1 | grep -E "passed|failed|^E " | tail -4
run() { echo "== $*"; timeout 600 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --c4-nvec 0 --sigma-sweep 0 --cpu-seconds 0 --build-from-host 0 --steps 20 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'emitted/q', r.get('rows_emitted_per_query'), 'serial', (d.get('serial') or {}).get('ms_per_step'), 'recall', d['recall_at_10'])" || tail -5 /tmp/err.txt; }
run --inflight 1 --opt screen16c_plane_seeds=0
run --inflight 1 --opt screen16c_plane_seeds=1
run --inflight 1 --opt screen16c_plane_seeds=0
run --inflight 1 --opt screen16c_plane_seeds=1
run --inflight 3
timeout 300 python3 tools/fuzz_scan.py 150 76 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
