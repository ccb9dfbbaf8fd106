#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $*"; timeout 600 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'serial', (d['serial'] or {}).get('ms_per_step'), (d['serial'] or {}).get('lanes_identical_to_serial'))" || tail -5 /tmp/err.txt; }
run --steps 40 --inflight 1
run --steps 40 --inflight 2
run --steps 40 --inflight 3
run --steps 40 --inflight 2 --opt screen16c_wave=3
run --steps 40 --inflight 3 --opt screen16c_wave=3
run --steps 40 --inflight 2 --opt screen16c_wave_blocks=3
run --steps 40 --inflight 3 --opt screen16c_wave_blocks=1
