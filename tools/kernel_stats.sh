#!/bin/bash
# Compact per-kernel resource table (VGPR / SGPR / scratch / LDS / occupancy) from hipcc's remarks.
# UNIT=<launch unit> picks the translation unit (default: launch_visualizer_strip; others: launch_generic, launch_visualizer_tiled, launch_separable, launch_resolve, capi, capi_readout, capi_audio)
cd "$(dirname "$0")/../shaderflow_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -fno-gpu-flush-denormals-to-zero -Wno-unused-value --cuda-device-only -c ${UNIT:-launch_visualizer_strip}.hip -o /tmp/unit.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
rows, cur = [], {}
for line in sys.stdin:
    m = re.search(r"remark:\s+(.+?): (\S+) \[-Rpass", line)
    if not m: continue
    k, v = m.group(1).strip(), m.group(2).strip()
    if k == "Function Name":
        cur = {"name": v}; rows.append(cur)
    else: cur[k] = v
names = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.split("\n")
pat = sys.argv[1] if len(sys.argv) > 1 else ""
print("%-70s %5s %5s %5s %8s %7s %4s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "LDS", "occ"))
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("sf::", "")
    if pat and pat not in n: continue
    print("%-70s %5s %5s %5s %8s %7s %4s" % (n[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$1"
