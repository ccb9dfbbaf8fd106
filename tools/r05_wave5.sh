#!/bin/bash
# round 5: three blocks a compute unit (two chunks in flight per wave) against two blocks (three chunks)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_ivf.py -x -q -m gpu 2>&1 | tail -4
run() { echo "== $NDBHIP_LIB $*"; timeout 600 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --steps 20 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'emitted/q', r.get('rows_emitted_per_query'))" || tail -5 /tmp/err.txt; grep trace /tmp/err.txt; }
run --opt screen16c_wave=0
run --opt screen16c_wave=3
run --opt screen16c_wave=2 --opt screen16c_wave_blocks=3
run --opt screen16c_wave=3
run --opt screen16c_wave=2 --opt screen16c_wave_blocks=3
run --opt screen16c_wave=2 --opt screen16c_wave_blocks=3 --strategy ip
run --opt screen16c_wave=2 --opt screen16c_wave_blocks=3 --dim 1536 --rows f16 --strategy ip --batch 256
run --opt screen16c_wave=0 --dim 1536 --rows f16 --strategy ip --batch 256
export NDBHIP_LIB=$GRAFT_REPO_ROOT/neurondb_amd/lib_ph/libndbhip.so
NDB_TRACE=gpurun_out/r05_wtrace_d3.npy run --opt screen16c_wave=3
NDB_TRACE=gpurun_out/r05_wtrace_d2b3.npy run --opt screen16c_wave=2 --opt screen16c_wave_blocks=3
unset NDBHIP_LIB
NDB_FUZZ_BLK3=1 timeout 300 python3 tools/fuzz_scan.py 150 74 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-300
