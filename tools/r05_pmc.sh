#!/bin/bash
# round 5: PMC passes over the register-streaming sweep (and its timing-only build without the pairs' requests)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp KERNELS="k_s16c_wsweep"
pass() { tag=$1; ctrs=$2; shift 2; timeout 400 bash tools/pmc_pass.sh $tag "$ctrs" "$@" > /dev/null 2>&1 </dev/null; echo "== $tag"; grep -A200 "PMC counters" gpurun_out/pmc_$tag.txt | grep "k_s16c_wsweep" | cut -c1-40,60-170; }
pass r05w_fetch "FETCH_SIZE" --opt screen16c_wave=3
pass r05w_write "WRITE_SIZE" --opt screen16c_wave=3
pass r05w_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --opt screen16c_wave=3
pass r05w_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --opt screen16c_wave=3
pass r05w_sq2 "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE" --opt screen16c_wave=3
pass r05w_tcp "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" --opt screen16c_wave=3
pass r05w_ta "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" --opt screen16c_wave=3
export NDBHIP_LIB=$GRAFT_REPO_ROOT/neurondb_amd/lib_nt/libndbhip.so
pass r05np_fetch "FETCH_SIZE" --opt screen16c_wave=3
pass r05np_tcc "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" --opt screen16c_wave=3
pass r05np_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" --opt screen16c_wave=3
