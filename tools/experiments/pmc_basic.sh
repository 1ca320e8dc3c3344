cd "$(dirname "$0")/../.." || exit 1
B="--scene basic --steps 2 --warmup 1 --no-cpu-baseline --no-export --frames-per-step 60 --camera-zoom 0.2"
tools/pmc_quick.sh "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" $B 2>&1 | grep -A12 "k_separable_fused"
tools/pmc_quick.sh "SQ_WAVES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM" $B 2>&1 | grep -A12 "k_separable_fused"
tools/pmc_quick.sh "SQ_WAVES SQ_INST_CYCLES_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F32" $B 2>&1 | grep -A12 "k_separable_fused"
tools/pmc_quick.sh "GRBM_GUI_ACTIVE GRBM_COUNT SQ_WAVES" $B 2>&1 | grep -A6 "k_separable_fused"
