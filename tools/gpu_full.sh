# full GPU suite + PMC of the bench kernel
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 1700 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02/pytest_gpu.txt
cat gpurun_out/r02/pytest_gpu.txt
bash tools/pmc_quick.sh "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" > gpurun_out/r02/pmc_a.txt 2>&1
bash tools/pmc_quick.sh "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_INSTS_VALU_TRANS" > gpurun_out/r02/pmc_b.txt 2>&1
cat gpurun_out/r02/pmc_a.txt gpurun_out/r02/pmc_b.txt
