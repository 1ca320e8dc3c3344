#!/bin/bash
# PC sampling of the benchmark's dominant kernel (rocprofv3's beta feature) on a GPU box: where the waves' program counters ARE, per
# instruction — the attribution of issue and wait cycles the PMC counters cannot give.
#   tools/pc_sample_bench.sh <tag> [stochastic|host_trap] [bench args…]
# Writes gpurun_out/pcs_<tag>/: head of the raw table, samples aggregated per (kernel, code-object offset[, columns the sampler adds]).
tag=${1:-r05}; method=${2:-stochastic}; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
out=$root/gpurun_out/pcs_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export ROCPROFILER_PC_SAMPLING_BETA_ENABLED=1
if [ "$method" = stochastic ]; then unit=cycles; interval=${PCS_INTERVAL:-1048576}; else unit=time; interval=${PCS_INTERVAL:-100}; fi
timeout ${PCS_TIMEOUT:-240} rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method $method --pc-sampling-unit $unit --pc-sampling-interval $interval \
    --kernel-trace -f csv -d /tmp/pcs_raw_$tag -o pcs -- python3 "$root/bench.py" --steps 2 --warmup 1 --frames-per-step 60 --no-cpu-baseline --no-export --no-live-counters "$@" \
    > "$out/bench.log" 2>&1
echo "rocprofv3 rc $?" >> "$out/bench.log"
find /tmp/pcs_raw_$tag -type f | head -20 > "$out/files.txt"
python3 "$root/tools/pc_samples_aggregate.py" /tmp/pcs_raw_$tag "$out" >> "$out/bench.log" 2>&1
tail -5 "$out/bench.log"
