#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2 3 4 5; do
timeout 900 python3 bench.py > gpurun_out/r04_bench_line_$i.json 2> gpurun_out/r04_bench_$i.log </dev/null
echo "run $i rc=$?"; grep -v amdgpu.ids gpurun_out/r04_bench_$i.log | tail -2
done
