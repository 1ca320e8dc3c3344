# round-2 evidence run on the GPU box: profile of the bench command (kernel trace + PMC passes), bench lines of the other
# configurations, micro-benchmarks, parity histogram. Everything lands in gpurun_out/r02/ (copied to profiles/ by hand).
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
bash tools/profile_bench.sh r02 > gpurun_out/r02/profile_summary_stdout.txt 2>&1
cp gpurun_out/prof_r02_summary.txt gpurun_out/r02/rocprofv3_bench_c3_summary.txt
cp gpurun_out/prof_r02.json gpurun_out/r02/bench_c3.json
cp gpurun_out/prof_r02.json profiles/r02_bench_c3.json      # the final bench line below reads its counters from here (same sources, same box)
timeout 120 build/ubench_valu > gpurun_out/r02/ubench_valu.txt 2>&1
timeout 120 build/ubench_mfma_valu > gpurun_out/r02/ubench_mfma_valu.txt 2>&1
python tools/parity_histogram.py > gpurun_out/r02/parity_histogram.txt 2>&1
: > gpurun_out/r02/bench_other_configs.jsonl
for cfg in "--width 256 --height 256 --ssaa 1" "--width 1920 --height 1080 --ssaa 1" "--width 1920 --height 1080 --ssaa 2" "--width 2560 --height 1440 --ssaa 2" \
           "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8" "--scene bars" "--scene bars --width 1920 --height 1080 --ssaa 2"; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $cfg 2>/dev/null | tail -1 >> gpurun_out/r02/bench_other_configs.jsonl
done
SHADERFLOW_VIS_FAST=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-export 2>/dev/null | tail -1 > gpurun_out/r02/bench_c3_round1_kernel.json
SHADERFLOW_SEPARABLE=0 timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-export --scene bars 2>/dev/null | tail -1 > gpurun_out/r02/bench_bars_round1_kernel.json
timeout 600 python bench.py > gpurun_out/r02/bench_c3.line.json 2> gpurun_out/r02/bench_c3.err
tail -2 gpurun_out/r02/bench_c3.err; cat gpurun_out/r02/bench_c3.line.json
python3 - <<'PY'
import json
for line in open("gpurun_out/r02/bench_other_configs.jsonl"):
    d = json.loads(line); print(d["metric"], d["value"], d["roofline"]["kernel"], d["roofline"]["launch_ms"], (d.get("export_host") or {}).get("value"))
PY
