#!/bin/bash
# GPU box: dynamic instruction counts of the rolled-camera kernel (and of the strip kernel beside it, whose per-supersample count the
# static census knows: the calibration) — SQ_INSTS_VALU / SQ_WAVES / SQ_ACTIVE_INST_VALU per dispatch. Lands in gpurun_out/rolled/.
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
export DEGREES=0,17,45
mkdir -p gpurun_out/rolled
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY -f csv -d gpurun_out/rolled/pmc -o pmc -- python3 tools/experiments/rotated_camera.py > gpurun_out/rolled/run.log 2>&1
python3 - <<'PY' | tee gpurun_out/rolled/counters.txt
import csv, glob
from collections import defaultdict
pmc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/rolled/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].replace("sf::", "").replace("void ", "").split("(")[0]
        pmc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
samples = 3840*2*2160*2
for k, c in pmc.items():
    if "isualizer" not in k: continue
    avg = {n: sum(v)/len(v) for n, v in c.items()}
    print(f"{k[:70]:70s} dispatches {len(c['SQ_WAVES'])}: waves {avg['SQ_WAVES']:.0f}, VALU wave-instructions {avg['SQ_INSTS_VALU']:.0f} = {avg['SQ_INSTS_VALU']*64/samples:.1f} per supersample, "
          f"SALU {avg['SQ_INSTS_SALU']*64/samples:.1f}, LDS {avg['SQ_INSTS_LDS']*64/samples:.1f} per supersample; VALU-active share of wave cycles {avg['SQ_ACTIVE_INST_VALU']/max(1,avg['SQ_WAVE_CYCLES']):.3f}, "
          f"busy cycles {avg['SQ_BUSY_CYCLES']:.0f}")
PY
