#!/bin/bash
# more configurations (crash / fallback check of the paths round 3 added): tools/r03_configs2.sh TAG
tag=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_other_configs2.txt
echo "# bench.py <args> --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 3 --build-from-host 0 --steps 10 : queries/s, ms/step, sweep ms, fallbacks/batches, rows rescored per query, swept/probed, recall@10, cpu parity" > $out
run() {
  timeout 900 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 3 --build-from-host 0 --steps 10 2>gpurun_out/${tag}_cfg2.log </dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; st = d.get('library_stats') or {}
print('$*', '|', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), str(st.get('screen16_fallbacks')) + '/' + str(st.get('screen16_batches')), r.get('rows_rescored_per_query'), round(st.get('rows_swept', 0) / max(1, st.get('rows_scored', 1)), 4), d.get('recall_at_10'), (d.get('cpu_baseline') or {}).get('gpu_parity_on_sample'))" >> $out 2>&1 || { echo "$* | FAILED" >> $out; tail -3 gpurun_out/${tag}_cfg2.log >> $out; }
}
run --strategy cosine
run --strategy cosine --rows f16 --dim 1536 --batch 4096
run --strategy l2 --rows f16 --dim 1536 --batch 4096
run --strategy ip --data gauss
run --strategy cosine --data gauss
run --strategy cosine --batch 256
run --strategy ip --batch 64
run --lists 256 --probes 16
run --dim 100 --lists 512
run --k 64
run --k 100
cat $out
