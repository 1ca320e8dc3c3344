#!/bin/bash
# PMC passes over one case of tools/bench_jit_tile.py with and without the LDS tile → gpurun_out/tile/pmc_jit_tile.txt
# (copied to profiles/r02_jit_tile_pmc.txt by hand). usage: tools/pmc_jit_tile.sh [TAPS:WIDTHxHEIGHT:SSAA]
cd "$(dirname "$0")/.." || exit 1
export TMPDIR=/tmp
CASE=${1:-4:3840x2160:2}
mkdir -p gpurun_out/tile
OUT=gpurun_out/tile/pmc_jit_tile.txt
echo "# tools/pmc_jit_tile.sh $CASE: sfx_jit_fused_* of a translated $CASE blur (taps per side:size:ssaa), per launch, tile off / on" > $OUT
python3 tools/bench_jit_tile.py --only $CASE:0 > /dev/null 2>&1       # compile both code objects outside the profiler
python3 tools/bench_jit_tile.py --only $CASE:1 > /dev/null 2>&1
for tile in 0 1; do
  for pass in "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VMEM_RD" \
              "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_SMEM"; do
    rm -rf gpurun_out/tile/pmc && mkdir -p gpurun_out/tile/pmc
    rocprofv3 --kernel-trace --pmc $pass -f csv -d gpurun_out/tile/pmc -o q -- python3 tools/bench_jit_tile.py --only $CASE:$tile > gpurun_out/tile/pmc/log.txt 2>&1
    python3 - $tile >> $OUT <<'PY'
import csv, glob, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob("gpurun_out/tile/pmc/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][-60:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "sfx_jit" not in k: continue
    print(f"tile {sys.argv[1]}  {k}")
    for n, v in sorted(cs.items()):
        print(f"   {n:24s} {sum(v)/len(v):18.0f}")
PY
  done
  rm -rf gpurun_out/tile/pmc && mkdir -p gpurun_out/tile/pmc
  rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/tile/pmc -o q -- python3 tools/bench_jit_tile.py --only $CASE:$tile > gpurun_out/tile/pmc/log.txt 2>&1
  python3 - $tile >> $OUT <<'PY'
import csv, glob, sys
for f in glob.glob("gpurun_out/tile/pmc/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sfx_jit" in r["Name"]:
            print(f"tile {sys.argv[1]}  {r['Name'].split('(')[0]}: {r['Calls']} launches, average {float(r['AverageNs'])/1e3:.1f} us")
PY
done
rm -rf gpurun_out/tile/pmc
cat $OUT
