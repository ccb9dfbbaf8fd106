#!/bin/bash
# PMC passes of the C5 configuration on one GPU (what bench.py's c5 leg runs): fetch, write, tcc
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
KERNELS="k_s16c_sweep\|k_s16_fin" PASSES="fetch write tcc" bash tools/pmc_all.sh r04c5 --nvec 10000000 --dim 1536 --rows f16 --strategy ip --batch 256 --lists 4096 --components 4096
