#!/bin/bash
# One rocprofv3 kernel-trace pass over bench.py -> gpurun_out/prof_TAG.txt (per-kernel count / total / avg / min / max).
# usage: tools/prof_pass.sh TAG [bench args...]
set -u
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 2 --cpu-seconds 0 --recall-queries 0 --hnsw-nvec 0 --gauss-steps 0 "$@" > /tmp/prof_$tag.json 2> /tmp/prof_$tag.log </dev/null
f=$(find /tmp/prof_$tag -name "*.db" | head -1)
if [ -z "$f" ]; then echo "no db for $tag"; grep -v amdgpu /tmp/prof_$tag.log | tail -8; exit 0; fi
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f 40 > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.txt </dev/null
cp /tmp/prof_$tag.json $GRAFT_REPO_ROOT/gpurun_out/prof_${tag}_bench_line.json
head -${PROF_LINES:-30} $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.txt | cut -c1-150
