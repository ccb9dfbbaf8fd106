#!/bin/bash
# tools/r03_fz.sh TAG SECONDS SEED...: the screen16 tests, then fuzz campaigns
tag=$1; secs=$2; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_screen16.py -x -q -m gpu > gpurun_out/${tag}_tests.log 2>&1
tail -3 gpurun_out/${tag}_tests.log
: > gpurun_out/${tag}_fuzz.txt
for seed in "$@"; do
  timeout $((secs + 120)) python3 tools/fuzz_scan.py $secs $seed 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-700 >> gpurun_out/${tag}_fuzz.txt
done
cat gpurun_out/${tag}_fuzz.txt
