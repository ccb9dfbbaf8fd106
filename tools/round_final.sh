#!/bin/bash
# What the driver runs at round end, timed — smoke, the default bench line — then the small-batch and latency files, the
# forced-dist C4 line at N = 1 and fuzz campaigns:  gpurun -- 'bash tools/round_final.sh r05 [fuzz seconds per seed]'
R=${1:-r06}; FZ=${2:-300}
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
t0=$(date +%s)
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
t1=$(date +%s); echo "smoke: $((t1-t0)) s"
timeout 1700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_bench_line.json 2> gpurun_out/${R}_bench.log </dev/null
t2=$(date +%s); echo "bench: $((t2-t1)) s"; tail -3 gpurun_out/${R}_bench.log | cut -c1-300
cp bench_detail.json gpurun_out/${R}_bench_detail.json
python3 - $R <<'PY'
import json, sys
R = sys.argv[1]
raw = open('gpurun_out/%s_bench_line.json' % R).read().strip().splitlines()[-1]
d = json.loads(raw)
print('line bytes', len(raw))
r = d['roofline']
print('value (i.i.d.)', d['value'], d['ms_per_step'], 'frac', r['frac'], r['bound'], r['kernel'], 'sweep', r.get('avg_launch_ms'), 'traffic', r.get('traffic'))
print('value_clustered', d.get('value_clustered'), d.get('ms_per_step_clustered'), (d.get('roofline_clustered') or {}).get('frac'))
print('cpu', d.get('cpu_baseline'))
for k in ('c4', 'c5', 'hnsw', 'balanced_index'):
    print(k, d.get(k))
for k, v in (d.get('sigma_sweep') or {}).items():
    print('  ', k, v)
print('build', d.get('build'))
print('config', d['config'])
PY
timeout 900 python3 bench.py --gpus 1 --force-dist --data clustered --nvec 10000000 --lists 4096 --shard slices --steps 10 --warmup 3 --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --c4-nvec 0 --sigma-sweep 0 --cpu-seconds 0 --build-from-host 0 2>/tmp/fd.err | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('forced-dist C4 at N=1:', d['value'], d['ms_per_step'], d['config']['collectives'], d.get('dist_parity_on_sample'))" || tail -5 /tmp/fd.err
NQS=1,8,16,32,64,128,256,512,1024 timeout 600 python3 tools/small_batch_probe.py 2>&1 | grep -E "nq=" > gpurun_out/${R}_small_batch.txt; cat gpurun_out/${R}_small_batch.txt
timeout 300 python3 tools/latency.py 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/${R}_latency.txt; cat gpurun_out/${R}_latency.txt
: > gpurun_out/${R}_fuzz_scan.txt
for seed in 71 72; do
  timeout $((FZ+120)) python3 tools/fuzz_scan.py $FZ $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700 >> gpurun_out/${R}_fuzz_scan.txt
done
timeout $((FZ/2+120)) python3 tools/fuzz_hnsw.py $((FZ/2)) 73 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-700 >> gpurun_out/${R}_fuzz_scan.txt
cat gpurun_out/${R}_fuzz_scan.txt
