#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench workload; results land in
# gpurun_out/prof_<tag>/ and are summarised into gpurun_out/prof_<tag>_summary.txt (copy what matters to profiles/).
# usage: tools/profile_bench.sh <tag> [bench args…]
set -u
TAG=${1:-r01}; shift || true
ARGS=${*:---steps 2 --warmup 1 --no-cpu-baseline --no-export}
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/prof_$TAG
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats -f csv -d "$OUT/trace" -o trace -- python3 bench.py $ARGS > "$OUT/bench_trace.log" 2>&1
for pass in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY" \
            "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VMEM" \
            "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_ADD_F16 SQ_INSTS_VALU_FMA_F16" \
            "SQ_INST_CYCLES_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU2 SQ_BUSY_CU_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT" \
            "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --pmc $pass -f csv -d "$OUT/pmc_$name" -o pmc -- python3 bench.py $ARGS > "$OUT/bench_pmc_$name.log" 2>&1
done
python3 tools/summarize_profile.py "$OUT" > "gpurun_out/prof_${TAG}_summary.txt" 2>&1
cat "gpurun_out/prof_${TAG}_summary.txt"
