#!/bin/bash
# On the GPU box: `rocprofv3 --kernel-trace --stats` of one short bench run per configuration beside C3 → the per-kernel table of each
# (calls, total, average, share) in gpurun_out/$TAG/other_configs_rocprofv3.txt — the traces the counters of
# profiles/$TAG_bench_other_configs.jsonl belong to.       usage: TAG=r06 tools/gpu_other_configs_trace.sh
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
TAG=${TAG:-r06}
mkdir -p gpurun_out/$TAG
OUT=gpurun_out/$TAG/other_configs_rocprofv3.txt
echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-live-counters <configuration>  (tools/gpu_other_configs_trace.sh)" > $OUT
for cfg in "--scene basic --width 256 --height 256 --ssaa 1 --frames-per-step 60" "--width 1920 --height 1080 --ssaa 1 --frames-per-step 60" "--width 1920 --height 1080 --ssaa 2 --frames-per-step 60" \
           "--width 2560 --height 1440 --ssaa 2 --frames-per-step 60" "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8" \
           "--scene bars --frames-per-step 300" "--scene waveform --frames-per-step 300" "--scene basic --frames-per-step 300"; do
  rm -rf /tmp/oc_trace
  rocprofv3 --kernel-trace --stats -f csv -d /tmp/oc_trace -o trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-export --no-live-counters $cfg > /tmp/oc_trace.log 2>&1
  echo >> $OUT; echo "## $cfg" >> $OUT
  python3 - >> $OUT <<'PY'
import csv, glob
for path in glob.glob("/tmp/oc_trace/**/*kernel_stats.csv", recursive=True):
    rows = list(csv.DictReader(open(path)))
    print(f"{'kernel':72s} {'calls':>6s} {'total ms':>10s} {'avg us':>12s} {'%':>6s}")
    for r in rows[:6]:
        name = r["Name"].replace("sf::", "").replace("void ", "").split("(")[0][:70]
        print(f"{name:72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:12.2f} {float(r['Percentage']):6.2f}")
PY
done
cat $OUT
