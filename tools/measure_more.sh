#!/bin/bash
# Secondary measurements of one state of the library (tools/measure_all.sh holds the judged line + profiles):
#   tools/measure_more.sh TAG  -> gpurun_out/TAG_other_configs.txt, TAG_service_bench.txt, TAG_small_batch.txt, TAG_fuzz_scan.txt
tag=$1
cd $GRAFT_REPO_ROOT
out=gpurun_out/${tag}_other_configs.txt
echo "# bench.py <args> --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --steps 10 : queries/s, ms/step, dominant-kernel ms/launch, screen16 fallbacks, rows rescored per query, build vectors/s, recall@10" > $out
run() {
  timeout 400 python3 bench.py "$@" --hnsw-nvec 0 --gauss-steps 0 --cpu-seconds 0 --steps 10 2>/dev/null </dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
r = d['roofline']; st = d.get('library_stats') or {}
print('$*', '|', d['value'], d['ms_per_step'], r.get('avg_launch_ms'), st.get('screen16_fallbacks'), r.get('rows_rescored_per_query'), d['build_vectors_per_s'], d.get('recall_at_10'))" >> $out
}
run --strategy ip
run --strategy cosine
run --batch 256
run --batch 1024
run --rows f16 --strategy ip --dim 1536 --batch 256
run --rows f16 --strategy ip --dim 1536 --batch 4096
run --nvec 10000000 --lists 4096 --components 4096
run --nvec 10000000 --lists 4096 --components 4096 --dim 1536 --rows f16 --strategy ip --batch 256
cat $out
sb=gpurun_out/${tag}_service_bench.txt; : > $sb
for cfg in "1 1 400" "16 1 400" "16 16 2000" "32 32 2000"; do set -- $cfg
  timeout 300 python3 tools/service_bench.py --backends $1 --inflight $2 --queries $3 --n 1000000 --dim 768 --nlists 1024 --nprobe 32 >> $sb 2>/dev/null </dev/null
done; cut -c1-200 $sb
timeout 300 python3 tools/small_batch_probe.py > gpurun_out/${tag}_small_batch.txt 2>/dev/null </dev/null; tail -12 gpurun_out/${tag}_small_batch.txt
if [ "${NOFUZZ:-0}" != "1" ]; then timeout 600 python3 tools/fuzz_scan.py 240 7 > gpurun_out/${tag}_fuzz_scan.txt 2>&1 </dev/null; tail -3 gpurun_out/${tag}_fuzz_scan.txt; fi
