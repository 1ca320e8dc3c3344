#!/bin/bash
# kernel_stats.sh with extra -D flags: tools/kernel_stats_extra.sh "<flags>" <kernel name pattern>
cd "$(dirname "$0")/../shaderflow_amd/csrc" || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -fno-slp-vectorize -fno-gpu-flush-denormals-to-zero -Wno-unused-value $1 --cuda-device-only -c ${UNIT:-launch_visualizer_strip}.hip -o /tmp/unit_x.o \
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
for r, n in zip(rows, names):
    n = re.sub(r"\(.*", "", n).replace("sf::", "")
    if pat and pat not in n: continue
    print("%-60s VGPR %4s SGPR %4s scratch %5s LDS %6s occ %2s" % (n[:60], r.get("VGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("LDS Size [bytes/block]"), r.get("Occupancy [waves/SIMD]")))
' "$2"
