# round-5 evidence run on the GPU box: profile of the bench command (kernel trace + PMC passes), the strip kernel's ISA census against
# that run, the filterbank as CSR and on MFMA, light kernels, bench lines of the other configurations, frame-loop scenes at kernel rate,
# rolled cameras, parity histogram, the final bench line. Everything lands in gpurun_out/r05/ (copied to profiles/ by hand).
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp TAG=r05
mkdir -p gpurun_out/r05
bash tools/profile_bench.sh r05 > gpurun_out/r05/profile_summary_stdout.txt 2>&1
cp gpurun_out/prof_r05_summary.txt gpurun_out/r05/rocprofv3_bench_c3_summary.txt
cp gpurun_out/prof_r05.json gpurun_out/r05/bench_c3.json
cp gpurun_out/prof_r05.json profiles/r05_bench_c3.json      # the final bench line below reads its counters from here (same sources, same box)
find gpurun_out/prof_r05/trace -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r05/rocprofv3_kernel_stats.csv
bash tools/experiments/filterbank_timing.sh > /dev/null 2>&1
bash tools/gpu_profile_light_r05.sh > /dev/null 2>&1
python tools/parity_histogram_r03.py > gpurun_out/r05/parity_histogram.txt 2>&1
: > gpurun_out/r05/bench_other_configs.jsonl
for cfg in "--width 256 --height 256 --ssaa 1 --frames-per-step 60" "--width 1920 --height 1080 --ssaa 1 --frames-per-step 60" "--width 1920 --height 1080 --ssaa 2 --frames-per-step 60" "--width 2560 --height 1440 --ssaa 2 --frames-per-step 60" \
           "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8" "--scene bars --frames-per-step 60" "--scene waveform --frames-per-step 60" "--scene basic --frames-per-step 60" "--scene bars --width 1920 --height 1080 --ssaa 2 --frames-per-step 60" \
           "--scene bars --frames-per-step 300" "--scene waveform --frames-per-step 300" "--scene basic --frames-per-step 300"; do
  timeout 300 python bench.py --steps 4 --warmup 2 --no-cpu-baseline $cfg 2>/dev/null | tail -1 >> gpurun_out/r05/bench_other_configs.jsonl
done
{ python tools/profile_frame_loop.py; python tools/profile_clock_loop.py; python tools/experiments/clock_scenes_rates.py; } 2>&1 | grep "frames/s" > gpurun_out/r05/frame_loop.txt
{ python tools/profile_export.py own; python tools/experiments/export_like_bench.py --first-leg; } 2>&1 | grep -E "frames/s" > gpurun_out/r05/export_stress.txt
{ python tools/experiments/rotated_camera.py; } 2>&1 | grep "camera rotated" > gpurun_out/r05/rolled_camera.txt
{ ZOOM_LIST="1 0.2 2.2 0.74" tools/experiments/basic_variants.sh; } 2>&1 | grep zoom > gpurun_out/r05/basic_tiers.txt
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/r05/bench_c3.line.json 2> gpurun_out/r05/bench_c3.err
tail -2 gpurun_out/r05/bench_c3.err; cat gpurun_out/r05/bench_c3.line.json
python3 - <<'PY'
import json
for line in open("gpurun_out/r05/bench_other_configs.jsonl"):
    d = json.loads(line); print(d["metric"], d["value"], d["roofline"]["kernel"], d["roofline"]["launch_ms"], d["roofline"].get("frames_per_launch"), d["roofline"].get("frac"), (d.get("export_host") or {}).get("value"))
PY
cat gpurun_out/r05/filterbank.txt
tail -30 gpurun_out/r05/parity_histogram.txt
