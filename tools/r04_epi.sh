#!/bin/bash
# round 4, first box: the sweep's pass 0 (matrix-pipe screen of the accumulator blocks): model tests, parity tests, the dense
# (i.i.d.) sweep with and without it and under the timing-only debug modes, then the clustered headline step both ways
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_mfma_model.py tests/test_gpu_screen16.py -x -q -m gpu > gpurun_out/r04a_tests.log 2>&1
tail -5 gpurun_out/r04a_tests.log
timeout 900 python3 tools/dense_probe.py "" "screen16c_epi=0" "screen16_debug=1" "screen16_debug=1,screen16c_epi=0" "screen16_debug=2" "screen16_debug=2,screen16c_epi=0" "screen16c_qb=4" "screen16c_qb=4,screen16c_epi=0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04a_dense.txt
DATA=clustered timeout 600 python3 tools/dense_probe.py "" "screen16c_epi=0" "" "screen16c_epi=0" 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04a_clustered.txt
