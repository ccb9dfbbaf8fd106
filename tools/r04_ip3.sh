#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_ivf.py tests/test_gpu_fullsize.py -x -q -m gpu > gpurun_out/r04_ip_tests.log 2>&1
tail -3 gpurun_out/r04_ip_tests.log
timeout 600 python3 tools/c5_check.py 1000000 1024 3 2>&1 | grep -v amdgpu | tail -2
run() { timeout 1200 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'), 'emitted', r.get('rows_emitted_per_query'), 'fallbacks', d['library_stats']['screen16_fallbacks'])" "$*"; }
run --dim 1536 --rows f16 --strategy ip --batch 256
run --nvec 10000000 --dim 1536 --rows f16 --strategy ip --batch 256 --lists 4096 --components 4096 --steps 20 --warmup 3
run --nvec 10000000 --lists 4096 --components 4096 --steps 20 --warmup 3
bash tools/r04_c5prof.sh 2>&1 | grep -E "seed|pair_|qcprep|finalize|k_s16c_sweep|cent_select|sub_pairs|k_s16_sweep<0, 0, 4, 2, 0, 3>"
