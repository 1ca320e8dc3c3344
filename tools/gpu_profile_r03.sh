# round-3 evidence run on the GPU box: profile of the bench command (kernel trace + PMC passes), light kernels (WRITE/FETCH per
# kernel), bench lines of the other configurations, parity histogram. Everything lands in gpurun_out/r03/ (copied to profiles/).
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r03
bash tools/profile_bench.sh r03 > gpurun_out/r03/profile_summary_stdout.txt 2>&1
cp gpurun_out/prof_r03_summary.txt gpurun_out/r03/rocprofv3_bench_c3_summary.txt
cp gpurun_out/prof_r03.json gpurun_out/r03/bench_c3.json
cp gpurun_out/prof_r03.json profiles/r03_bench_c3.json      # the final bench line below reads its counters from here (same sources, same box)
find gpurun_out/prof_r03/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r03/rocprofv3_kernel_stats.csv
bash tools/gpu_profile_light_r03.sh > /dev/null 2>&1
python tools/parity_histogram_r03.py > gpurun_out/r03/parity_histogram.txt 2>&1
: > gpurun_out/r03/bench_other_configs.jsonl
for cfg in "--width 256 --height 256 --ssaa 1" "--width 1920 --height 1080 --ssaa 1" "--width 1920 --height 1080 --ssaa 2" "--width 2560 --height 1440 --ssaa 2" \
           "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8" "--scene bars" "--scene waveform" "--scene basic" "--scene bars --width 1920 --height 1080 --ssaa 2"; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $cfg 2>/dev/null | tail -1 >> gpurun_out/r03/bench_other_configs.jsonl
done
timeout 900 python bench.py > gpurun_out/r03/bench_c3.line.json 2> gpurun_out/r03/bench_c3.err
tail -2 gpurun_out/r03/bench_c3.err; cat gpurun_out/r03/bench_c3.line.json
python3 - <<'PY'
import json
for line in open("gpurun_out/r03/bench_other_configs.jsonl"):
    d = json.loads(line); print(d["metric"], d["value"], d["roofline"]["kernel"], d["roofline"]["launch_ms"], (d.get("export_host") or {}).get("value"))
PY
tail -30 gpurun_out/r03/parity_histogram.txt
