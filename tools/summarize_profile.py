#!/usr/bin/env python3
"""Summarises a tools/profile_bench.sh output directory: per-kernel time (kernel-trace stats) and PMC counters
(mean per dispatch) of the dominant kernels. Plain text, meant to be committed under profiles/."""
import csv
import sys
from collections import defaultdict
from pathlib import Path

root = Path(sys.argv[1])


def short(name: str) -> str:
    name = name.replace("sf::", "").replace("void ", "")
    return name.split("(")[0][:70]


print(f"# profile summary of {root}")
for stats in sorted(root.glob("trace/**/*kernel_stats.csv")):
    print(f"\n## kernel-trace --stats ({stats.name})")
    with open(stats) as fh:
        rows = list(csv.DictReader(fh))
    print(f"{'kernel':72s} {'calls':>6s} {'total ms':>10s} {'avg us':>12s} {'%':>6s}")
    for r in rows[:14]:
        print(f"{short(r['Name']):72s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:10.3f} {float(r['AverageNs'])/1e3:12.2f} {float(r['Percentage']):6.2f}")

for log in sorted(root.glob("bench_trace.log")):
    lines = [l for l in log.read_text().splitlines() if l.startswith("{")]
    if lines:
        print("\n## bench line under the tracer\n" + lines[-1])

print("\n## PMC counters (mean per dispatch of each kernel; separate passes)")
acc = defaultdict(lambda: defaultdict(list))
for counters in sorted(root.glob("pmc_*/**/*counter_collection.csv")):
    with open(counters) as fh:
        for r in csv.DictReader(fh):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for kernel, cs in acc.items():
    if not any(word in kernel for word in ("render", "visualizer", "resolve", "stft", "dynamics", "filterbank", "separable")):
        continue
    print(f"\n{kernel}")
    for name, values in sorted(cs.items()):
        print(f"  {name:28s} mean {sum(values)/len(values):18.1f}   dispatches {len(values)}")


# machine-readable twin (bench.py reads the dominant kernel's counters from it): <root>.json next to the directory
import json
import subprocess
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from shaderflow_amd._native import source_fingerprint                # noqa: E402
durations = {}
for stats in sorted(root.glob("trace/**/*kernel_stats.csv")):
    with open(stats) as fh:
        for r in csv.DictReader(fh):
            durations[short(r["Name"])] = {"calls": int(r["Calls"]), "average_ns": float(r["AverageNs"]), "percentage": float(r["Percentage"])}
bench_line = None
for log in sorted(root.glob("bench_trace.log")):
    lines = [l for l in log.read_text().splitlines() if l.startswith("{")]
    bench_line = json.loads(lines[-1]) if lines else None
try:
    head = subprocess.run(["git", "rev-parse", "HEAD"], capture_output=True, text=True, cwd=Path(__file__).resolve().parent.parent).stdout.strip()
except OSError:
    head = ""
record = {"source_fingerprint": source_fingerprint(), "git_head": head, "command": " ".join(sys.argv[2:]),
          "kernels": {kernel: {"duration": durations.get(kernel), "counters": {name: sum(v)/len(v) for name, v in cs.items()}}
                      for kernel, cs in acc.items()},
          "durations": durations, "bench_under_tracer": bench_line}
Path(str(root).rstrip("/") + ".json").write_text(json.dumps(record, indent=1))
