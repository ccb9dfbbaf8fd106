#!/bin/bash
# One rocprofv3 PMC pass over bench.py (counters in their own run: --kernel-trace only, no other tracing).
# usage: tools/pmc_pass.sh TAG "COUNTER [COUNTER ...]" [bench args...]   -> gpurun_out/pmc_TAG.txt
# PROG="tools/h2_bench.py 1000000 768 clustered 64" tools/pmc_pass.sh ... profiles that program instead of bench.py
set -u
tag=$1; shift
ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$tag
if [ -n "${PROG:-}" ]; then
rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/$PROG > /tmp/pmc_$tag.log 2>&1
else
rocprofv3 --pmc $ctrs --kernel-trace -d /tmp/pmc_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --cpu-seconds 0 --recall-queries 0 --hnsw-nvec 0 --gauss-steps 0 --c5-nvec 0 --c4-nvec 0 --sigma-sweep 0 --inflight 1 --build-from-host 0 "$@" > /tmp/pmc_$tag.log 2>&1
fi
f=$(find /tmp/pmc_$tag -name "*.db" | head -1)
if [ -z "$f" ]; then echo "no db for $tag"; grep -v amdgpu /tmp/pmc_$tag.log | tail -8; exit 0; fi
python3 $GRAFT_REPO_ROOT/tools/rocpd_summary.py $f > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt
grep -A40 "PMC counters" $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.txt | grep -i "grouped\|counter" | head -20
