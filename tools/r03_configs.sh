#!/bin/bash
# the other configurations with round 3's library: tools/r03_configs.sh TAG -> gpurun_out/TAG_other_configs.txt
tag=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_other_configs.txt
echo "# bench.py <args> --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --steps 10 : queries/s, ms/step, dominant-kernel ms/launch, screen16 fallbacks, rows rescored per query, rows swept / rows probed, build vectors/s, prepare s, recall@10" > $out
run() {
  timeout 900 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --build-from-host 0 --steps 10 2>gpurun_out/${tag}_cfg.log </dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d['roofline']; st = d.get('library_stats') or {}; b = d.get('build') or {}
print('$*', '|', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), st.get('screen16_fallbacks'), r.get('rows_rescored_per_query'), round(st.get('rows_swept', 0) / max(1, st.get('rows_scored', 1)), 4), d['build_vectors_per_s'], b.get('prepare_seconds'), d.get('recall_at_10'))" >> $out 2>&1 || { echo "$* | FAILED" >> $out; tail -3 gpurun_out/${tag}_cfg.log >> $out; }
}
for a in "$@"; do :; done
run --strategy ip
run --strategy cosine
run --batch 256
run --batch 1024
run --rows f16 --strategy ip --dim 1536 --batch 256
run --rows f16 --strategy ip --dim 1536 --batch 4096
run --nvec 10000000 --lists 4096 --components 4096
run --nvec 10000000 --lists 4096 --components 4096 --dim 1536 --rows f16 --strategy ip --batch 256
run --nvec 10000000 --lists 4096 --components 4096 --dim 1536 --rows f16 --strategy ip --batch 4096
cat $out
