#!/bin/bash
# k_separable_fused<default>'s own duration per frame under the cameras of basic_tiers.py (rocprofv3 kernel trace). GPU box only.
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
for zoom in ${ZOOM_LIST:-1 0.2 2.2 0.74}; do
  out=gpurun_out/basic_tiers/z$zoom
  mkdir -p "$out"
  ZOOMS=$zoom LAUNCHES=200 rocprofv3 --kernel-trace --stats -f csv -d "$out" -o t -- python3 tools/experiments/basic_tiers.py > "$out/log.txt" 2>&1
  tail -1 "$out/log.txt"
  stats=$(find "$out" -name "t_kernel_stats.csv" | head -1)
  python3 - "$stats" <<'PY'
import csv, sys
for row in csv.DictReader(open(sys.argv[1])):
    if row["Name"].startswith(("void sf::k_separable", "sf::k_separable", "k_separable")) or "k_separable" in row["Name"]:
        print(f'    {row["Name"][:60]:60s} calls {row["Calls"]:>5s}  avg {float(row["AverageNs"])/1e3:8.2f} us')
PY
done
