#!/bin/bash
# round 5: the register-streaming sweep (k_s16c_wsweep) against the LDS ring, parity first
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_ivf.py -x -q -m gpu 2>&1 | tail -8
run() { echo "== $*"; timeout 600 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 --steps 20 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('q/s', d['value'], 'ms/step', d['ms_per_step'], 'sweep ms', r.get('avg_launch_ms'), 'frac', r.get('frac'), 'bytes', r.get('hbm',{}).get('bytes_per_launch'), 'emitted/q', r.get('rows_emitted_per_query'), 'rescored/q', r.get('rows_rescored_per_query'))" || tail -5 /tmp/err.txt; }
run --opt screen16c_wave=0
run --opt screen16c_wave=5
run --opt screen16c_wave=4
run --opt screen16c_wave=3
run --opt screen16c_wave=2
run --strategy ip --opt screen16c_wave=0
run --strategy ip --opt screen16c_wave=5
for seed in 71; do
  timeout 420 python3 tools/fuzz_scan.py 200 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-600
done
