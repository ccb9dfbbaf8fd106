#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu -k "streamed or centroid_scan or matches_oracle" 2>&1 | tail -5
run() { timeout 1200 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(sys.argv[1], d['value'], d['ms_per_step'], r.get('avg_launch_ms'), 'rescored', r.get('rows_rescored_per_query'), 'emitted', r.get('rows_emitted_per_query'), 'fallbacks', d['library_stats']['screen16_fallbacks'])" "$*"; }
run --steps 30
run --strategy ip --steps 30
run --dim 1536 --rows f16 --strategy ip --batch 256
rm -rf /tmp/ks2; export TMPDIR=/tmp
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/ks2 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --recall-queries 0 > /tmp/ks2.log 2>&1)
f=$(find /tmp/ks2 -name "*.db" | head -1)
[ -n "$f" ] && python3 tools/rocpd_summary.py $f 70 > gpurun_out/r04_c2_kernel_stats2.txt
grep -E "finalize|cent_select|seed|qcprep|sub_pairs|k_s16c_sweep" gpurun_out/r04_c2_kernel_stats2.txt | cut -c1-140
bash tools/r04_c5prof.sh 2>&1 | grep -E "seed|qcprep|finalize|k_s16c_sweep|cent_select|sub_pairs"
