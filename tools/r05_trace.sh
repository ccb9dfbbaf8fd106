#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
run() { echo "== $NDBHIP_LIB $*"; timeout 600 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --steps 20 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'emitted/q', r.get('rows_emitted_per_query'))" || tail -5 /tmp/err.txt; grep trace /tmp/err.txt; }
export NDBHIP_LIB=$GRAFT_REPO_ROOT/neurondb_amd/lib_ph/libndbhip.so
NDB_TRACE=gpurun_out/r05_wtrace_d3.npy run --opt screen16c_wave=3
