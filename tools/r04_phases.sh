#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export NDB_PHASES=1 TMPDIR=/tmp
timeout 600 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu -k "streamed" 2>&1 | tail -15
run() { echo "== $*"; timeout 1200 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 2>/tmp/err.txt | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"; grep phases /tmp/err.txt; }
run --dim 1536 --rows f16 --strategy ip --batch 256 --opt screen16_stage=0
run --dim 1536 --rows f16 --strategy ip --batch 256
run --steps 20 --opt screen16_stage=0
run --steps 20
prof() { rm -rf /tmp/ks3; (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks3 -o p -- python3 $GRAFT_REPO_ROOT/bench.py "$@" --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 > /tmp/ks3.log 2>&1)
f=$(find /tmp/ks3 -name "*.db" | head -1); [ -n "$f" ] && python3 tools/rocpd_summary.py $f 70 | grep -E "finalize|cent_select" | cut -c1-140; }
prof --steps 20
prof --dim 1536 --rows f16 --strategy ip --batch 256 --steps 20
