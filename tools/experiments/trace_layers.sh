#!/bin/bash
# per-LAYER kernel times of the layered scenes at 1080p: rocprofv3's per-name averages mix layer 0 and layer 1 of one kernel name, so the
# dispatches of a name are split by their order (even = layer 0, odd = layer 1) → gpurun_out/trace_layers.txt
export TMPDIR=/tmp
out=${1:-gpurun_out/trace_layers.txt}; : > $out
for scene in MotionBlur Multipass; do
  rm -rf /tmp/trace_layers_$scene
  rocprofv3 --kernel-trace -f csv -d /tmp/trace_layers_$scene -o t -- python3 tools/profile_frame_loop.py $scene 2>&1 | grep "frames/s" >> $out
  python3 - "$scene" /tmp/trace_layers_$scene >> $out <<'PY'
import csv, sys, glob, collections
scene, root = sys.argv[1], sys.argv[2]
rows = []
for path in glob.glob(root + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
seen = collections.Counter(); by = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].replace("sf::", "").replace("void ", "").split("(")[0][:70]
    by[(name, seen[name] % 2)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    seen[name] += 1
print(f"# {scene}: kernel, dispatch parity (0 = first of a frame), launches, average us")
for (name, parity), d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
    if len(d) > 50: print(f"  {name:70s} {parity} {len(d):6d} {sum(d)/len(d)/1e3:9.2f}")
PY
done
cat $out
