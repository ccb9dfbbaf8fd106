#!/bin/bash
# tools/r03_sb.sh TAG: small-batch latencies under a few option settings (one box, so the lines compare)
tag=$1
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for o in "" "cent_screen16=0" "screen16c_seeds=64" "screen16_sublists=0"; do
OPTS=$o MINNQS=32 NQS=64,128,256,512,1024 timeout 600 python3 tools/small_batch_probe.py > gpurun_out/${tag}_sb.log 2>&1
grep -E "nq=|screen_min|option" gpurun_out/${tag}_sb.log
done
