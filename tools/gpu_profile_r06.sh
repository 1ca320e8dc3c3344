# round-6 evidence run on the GPU box: profile of the bench command (kernel trace + PMC passes), one bench line per other configuration
# with the live counters of ITS dominant kernel, the tier's fall-back count, the parity histogram, the final bench line (the driver's
# command). Everything lands in gpurun_out/r06/ (what is judged is copied to profiles/ by hand, tools/census.sh runs against the
# line afterwards — it needs no GPU).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp TAG=r06
mkdir -p gpurun_out/r06
bash tools/profile_bench.sh r06 > gpurun_out/r06/profile_summary_stdout.txt 2>&1
cp gpurun_out/prof_r06_summary.txt gpurun_out/r06/rocprofv3_bench_c3_summary.txt
cp gpurun_out/prof_r06.json gpurun_out/r06/bench_c3.json; cp gpurun_out/prof_r06.json profiles/r06_bench_c3.json      # the final bench line below reads what its live passes do not give from here (same sources, same box)
find gpurun_out/prof_r06/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r06/rocprofv3_kernel_stats.csv
SHADERFLOW_BENCH_TILE_MISSES=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-live-counters 2>&1 | grep "tile misses" > gpurun_out/r06/tier_fallbacks.txt
for t in 0 1 0 1; do SHADERFLOW_VIS_PIXEL_TIER=$t python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-export --no-live-counters 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('SHADERFLOW_VIS_PIXEL_TIER=$t', d['value'], 'frames/s', d['roofline']['launch_ms'], 'ms per 300 frames')"; done > gpurun_out/r06/tier_ab.txt 2>&1
bash tools/gpu_other_configs.sh > gpurun_out/r06/other_configs_summary.txt 2>&1
python tools/parity_histogram_r03.py > gpurun_out/r06/parity_histogram.txt 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r06/bench_c3.line.json 2> gpurun_out/r06/bench_c3.err
tail -2 gpurun_out/r06/bench_c3.err; cut -c1-400 gpurun_out/r06/bench_c3.line.json; echo
cat gpurun_out/r06/tier_fallbacks.txt gpurun_out/r06/tier_ab.txt gpurun_out/r06/other_configs_summary.txt
tail -30 gpurun_out/r06/parity_histogram.txt
