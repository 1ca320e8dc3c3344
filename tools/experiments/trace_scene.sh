# kernel-trace of one bench scene: per-kernel average durations (usage: trace_scene.sh <scene args...>)
cd /root/repo; export TMPDIR=/tmp
rm -rf gpurun_out/trace_scene; mkdir -p gpurun_out/trace_scene
rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/trace_scene -o t -- python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-export "$@" > gpurun_out/trace_scene/bench.log 2>&1
python3 - <<'PY'
import csv, glob, json
for f in glob.glob("gpurun_out/trace_scene/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:9]:
        print(f"{r['Name'][:70]:70s} {r['Calls']:>5s} {float(r['AverageNs'])/1e3:10.1f} us {r['Percentage']:>6s} %")
line = [l for l in open("gpurun_out/trace_scene/bench.log") if l.startswith("{")]
if line: d = json.loads(line[-1]); print(d["value"], d["ms_per_step"])
PY
