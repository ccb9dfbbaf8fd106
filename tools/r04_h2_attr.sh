#!/bin/bash
# what the batched build schedule costs in recall: the same table and level draws with batch_max = 1
# (every insert sees every earlier one) next to the default schedule, i.i.d. unit rows and the clustered unit table
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for kind in gauss clustered; do
for bmax in 8192 256 1; do
echo "== $kind 100000 x 768, batch_max $bmax"
H2_BMAX=$bmax timeout 1500 python3 tools/h2_bench.py 100000 768 $kind 64 256 1024 2>&1 | grep -v amdgpu
done
done
} | tee gpurun_out/r04_h2_schedule.txt
bash tools/r04_prof_c2.sh
