# PMC passes of the bench kernel (10-frame launch); usage: tools/gpu_pmc.sh <tag> [env assignments…]
cd /root/repo
export TMPDIR=/tmp
TAG=${1:-x}
mkdir -p gpurun_out/r02
for pass in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
            "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
            "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  bash tools/pmc_quick.sh "$pass" 2>&1 | grep -A12 "visualizer_fast\|visualizer_strip\|VisualizerShader" | head -14
done > gpurun_out/r02/pmc_$TAG.txt 2>&1
cat gpurun_out/r02/pmc_$TAG.txt
