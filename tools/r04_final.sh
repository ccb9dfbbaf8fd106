#!/bin/bash
# what the driver runs at round end: smoke, the default bench line (timed), plus the small-batch and latency files
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
t0=$(date +%s)
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t1=$(date +%s); echo "smoke: $((t1-t0)) s"
timeout 1500 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r04_final_bench_line.json 2> gpurun_out/r04_final_bench.log </dev/null
t2=$(date +%s); echo "bench: $((t2-t1)) s"
tail -2 gpurun_out/r04_final_bench.log | cut -c1-300
python3 - <<PY
import json
d=json.loads(open('gpurun_out/r04_final_bench_line.json').read().strip().splitlines()[-1])
r=d['roofline']
print('C2', d['value'], d['ms_per_step'], 'frac', r['frac'], r['bound'], 'traffic', r.get('traffic'), 'step_frac', r['hbm']['step_frac'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'], d['cpu_baseline'].get('gpu_parity_on_sample'))
g=d['iid_gauss']; print('iid', g['queries_per_s'], g['ms_per_step'], g['roofline']['frac'], g['roofline']['mfma'].get('busy_pmc'), g['cpu_baseline']['value'] if g.get('cpu_baseline') else None)
b=d['build']; print('build', b['vectors_per_s'], b['searchable_vectors_per_s'], b['from_host_vectors_per_s'], (b['from_host'] or {}).get('roofline',{}).get('frac'))
h=d.get('hnsw') or {}; print('hnsw', {k:h.get(k) for k in ('queries_per_s','build_vectors_per_s')}, (h.get('roofline') or {}).get('frac'))
c5=d.get('c5') or {}; print('c5', {k:c5.get(k) for k in ('queries_per_s','ms_per_step','error')}, c5.get('exact_scan_parity'), c5.get('oracle_parity'))
print('config', d['config'])
PY
NQS=1,8,16,32,64,128,256,512,1024 timeout 600 python3 tools/small_batch_probe.py 2>&1 | grep -E "nq=" > gpurun_out/r04_final_small_batch.txt; cat gpurun_out/r04_final_small_batch.txt
timeout 300 python3 tools/latency.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r04_final_latency.txt; cat gpurun_out/r04_final_latency.txt
: > gpurun_out/r04_fuzz_scan.txt
for seed in 61 62; do
  timeout 420 python3 tools/fuzz_scan.py 300 $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700 >> gpurun_out/r04_fuzz_scan.txt
done
cat gpurun_out/r04_fuzz_scan.txt
