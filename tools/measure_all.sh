#!/bin/bash
# The judged measurements of one state of the library, in one call on the GPU box:  tools/measure_all.sh TAG
#   gpurun_out/TAG_bench_line.json      the default bench.py line
#   gpurun_out/prof_TAG.txt             rocprofv3 --kernel-trace --stats summary of the same command (shorter run)
#   gpurun_out/pmc_TAG_*.txt            PMC passes (tools/pmc_all.sh), gpurun_out/pmc_TAG_hnsw_*.txt (tools/pmc_hnsw.sh)
tag=$1
cd $GRAFT_REPO_ROOT
timeout 900 python3 bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench.log </dev/null
tail -c 600 gpurun_out/${tag}_bench_line.json; echo
timeout 900 bash tools/prof_pass.sh $tag </dev/null | head -24
timeout 1600 bash tools/pmc_all.sh $tag </dev/null
timeout 1300 bash tools/pmc_hnsw.sh $tag </dev/null
