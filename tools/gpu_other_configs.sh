#!/bin/bash
# On the GPU box: one bench line per configuration beside C3 — every one with the live counters of ITS dominant kernel (roofline.frac,
# valu_instructions_per_supersample, lds_busy, traffic: VERDICT round 5, missing 3) — into gpurun_out/$TAG/bench_other_configs.jsonl,
# and a one-line summary of each on stdout. (C1 = BASELINE's configs[0]: the Basic scene at 256 x 256 without SSAA; the Visualizer at that
# size beside it.)        usage: TAG=r06 tools/gpu_other_configs.sh
cd "$(dirname "$0")/.." || exit 1
TAG=${TAG:-r06}
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/bench_other_configs.jsonl
: > $OUT
for cfg in "--scene basic --width 256 --height 256 --ssaa 1 --frames-per-step 60" "--width 256 --height 256 --ssaa 1 --frames-per-step 60" "--width 1920 --height 1080 --ssaa 1 --frames-per-step 60" "--width 1920 --height 1080 --ssaa 2 --frames-per-step 60" \
           "--width 2560 --height 1440 --ssaa 2 --frames-per-step 60" "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8" \
           "--scene bars --frames-per-step 300" "--scene waveform --frames-per-step 300" "--scene basic --frames-per-step 300" "--scene basic --frames-per-step 300 --camera-zoom 0.2"; do
  timeout 600 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-export $cfg 2>>gpurun_out/$TAG/bench_other_configs.err | tail -n 1 >> $OUT
done
python3 - "$OUT" <<'PY'
import json, sys
for line in open(sys.argv[1]):
    if not line.strip(): continue
    d = json.loads(line); r = d["roofline"]
    print(f'{d["metric"]:58s} {d["value"]:>10.1f} fps  {r["kernel"][:52]:52s} share {r.get("kernel_share_of_gpu_time")}  {r["bound"]} frac {r.get("frac")}  instr/sample {r.get("valu_instructions_per_supersample")}  lds {r.get("lds_busy")}  traffic {r.get("traffic")}')
PY
