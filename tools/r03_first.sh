#!/bin/bash
# first GPU contact of the centred sweep: the screened-scan parity tests, a short fuzz campaign, the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_screen16.py tests/test_gpu_ivf.py tests/test_gpu_mfma_model.py -x -q -m gpu > gpurun_out/r03a_tests.log 2>&1
tail -15 gpurun_out/r03a_tests.log
timeout 300 python3 tools/fuzz_scan.py 120 7 > gpurun_out/r03a_fuzz.log 2>&1
tail -5 gpurun_out/r03a_fuzz.log
timeout 900 python3 bench.py --hnsw-nvec 0 > gpurun_out/r03a_bench_line.json 2> gpurun_out/r03a_bench.log </dev/null
tail -c 1500 gpurun_out/r03a_bench_line.json; echo
tail -5 gpurun_out/r03a_bench.log
