# Light fragments and the two-pass configuration: kernel-trace + WRITE_SIZE / FETCH_SIZE passes, achieved HBM GB/s per kernel.
# Runs on the GPU box; writes gpurun_out/${TAG}/light_kernels.txt (copied to profiles/${TAG}_light_kernels.txt by hand).
cd "$(dirname "$0")/.." || exit 1
TAG=${TAG:-r05}
export TMPDIR=/tmp
mkdir -p gpurun_out/${TAG}
OUT=gpurun_out/${TAG}/light_kernels.txt
: > $OUT
run() {  # tag, bench args…
  tag=$1; shift
  rm -rf gpurun_out/light_$tag; mkdir -p gpurun_out/light_$tag
  rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/light_$tag/trace -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --frames-per-step 60 "$@" > gpurun_out/light_$tag/bench.log 2>&1
  for c in WRITE_SIZE FETCH_SIZE; do
    rocprofv3 --kernel-trace --pmc $c -f csv -d gpurun_out/light_$tag/pmc_$c -o pmc -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export --frames-per-step 60 "$@" > gpurun_out/light_$tag/bench_$c.log 2>&1
  done
  python3 - "$tag" "$*" >> $OUT <<'PY'
import csv, glob, json, sys
from collections import defaultdict
tag, args = sys.argv[1], sys.argv[2]
short = lambda n: n.replace("sf::", "").replace("void ", "").split("(")[0][:64]
dur = {}
for f in glob.glob(f"gpurun_out/light_{tag}/trace/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[short(r["Name"])] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(f"gpurun_out/light_{tag}/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
line = [l for l in open(f"gpurun_out/light_{tag}/bench.log") if l.startswith("{")]
d = json.loads(line[-1]) if line else {}
print(f"## bench.py {args}: {d.get('value')} frames/s, {d.get('roofline', {}).get('frames_per_launch')} frames per launch")
print(f"{'kernel':66s} {'calls':>5s} {'avg us':>10s} {'% time':>7s} {'written MB':>11s} {'fetched MB':>11s} {'write GB/s':>11s} {'w+f GB/s':>9s} {'of 8 TB/s':>9s}")
for k, (calls, avg, pct) in sorted(dur.items(), key=lambda kv: -kv[1][2])[:6]:
    w = pmc[k].get("WRITE_SIZE"); f = pmc[k].get("FETCH_SIZE")
    if not w or not f: continue
    wb, fb = sum(w)/len(w)*1024, sum(f)/len(f)*1024           # KiB per dispatch
    print(f"{k:66s} {calls:5d} {avg/1e3:10.1f} {pct:7.2f} {wb/1e6:11.1f} {fb/1e6:11.1f} {wb/avg:11.1f} {(wb+fb)/avg:9.1f} {(wb+fb)/avg/8000:9.3f}")
print()
PY
}
run bars_c3 --scene bars
run waveform_c3 --scene waveform
run basic_c3 --scene basic
run visualizer_c2 --width 1920 --height 1080 --ssaa 1
cat $OUT
