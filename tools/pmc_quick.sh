#!/bin/bash
# One PMC pass over a short bench run; prints per-kernel means. usage: tools/pmc_quick.sh "SQ_INSTS_VALU SQ_WAVES" [bench args]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
C=${1:-SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU SQ_INSTS_LDS}; shift || true
ARGS=${*:---steps 1 --warmup 1 --no-cpu-baseline --no-export --frames-per-step 10}
rm -rf gpurun_out/pmcq && mkdir -p gpurun_out/pmcq
rocprofv3 --kernel-trace --pmc $C -f csv -d gpurun_out/pmcq -o q -- python3 bench.py $ARGS > gpurun_out/pmcq/log.txt 2>&1
python3 - <<'PY'
import csv, glob
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/pmcq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-90:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if not any(s in k for s in ("render", "visualizer", "separable", "resolve")): continue
    waves = sum(cs.get("SQ_WAVES", [1]))/max(1, len(cs.get("SQ_WAVES", [1])))
    print(k)
    for n, v in sorted(cs.items()):
        m = sum(v)/len(v)
        print(f"   {n:24s} {m:18.0f}   per wave {m/waves:10.1f}")
PY
