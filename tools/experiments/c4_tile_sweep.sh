#!/bin/bash
# BASELINE config 4's LDS-tile-size sweep on the strip kernel's 4x instance (missing 3): builds one variant of the strip
# unit per block geometry — COLUMN_GROUPS x WALK decide the block's output pixels (64*CG/4 wide, (8/CG)*WALK/4 tall) and, with the
# window bound, its LDS tile — here; benches each at 7680x4320 4xSSAA on the GPU box (`tools/experiments/c4_tile_sweep.sh bench`).
cd "$(dirname "$0")/../.." || exit 1
if [ "$1" = "bench" ]; then
  for so in shaderflow_amd/libshaderflow_hip.so build/variants/lib_c4_*.so; do
    SHADERFLOW_HIP_LIBRARY=$PWD/$so python3 bench.py --width 7680 --height 4320 --ssaa 4 --frames-per-step 8 --steps 4 --warmup 2 --no-cpu-baseline --no-export --no-live-counters 2>/dev/null | tail -n 1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-34s %7.1f frames/s  %8.3f ms per 8 frames  %s' % ('$so'.split('/')[-1], d['value'], d['roofline']['launch_ms'], d['roofline']['kernel']))"
  done
  # the same frame through every geometry: within 1 LSB of the shipped one's (the classification's tiles differ, nothing else does)
  python3 tools/experiments/c4_tile_sweep_check.py /tmp/c4_ref.npy > /dev/null
  for so in build/variants/lib_c4_*.so; do
    SHADERFLOW_HIP_LIBRARY=$PWD/$so python3 tools/experiments/c4_tile_sweep_check.py /tmp/c4_var.npy | tail -n 1 | tr '\n' ' '
    python3 -c "
import numpy as np
a, b = np.load('/tmp/c4_ref.npy').astype(int), np.load('/tmp/c4_var.npy').astype(int)
d = np.abs(a - b); print('vs the shipped geometry: max |d| =', d.max(), ' identical %.3f %%' % ((d == 0).mean()*100))"
  done
  exit 0
fi
rm -f build/variants/lib_c4_*.so
# name: pixels per block (wide x tall) — CG, WALK, tile pitch x rows (cells), waves per SIMD asked for
tools/variants.sh "c4_16x8:-DVIS4_SWEEP_CG=1 -DVIS4_SWEEP_WALK=4 -DVIS4_SWEEP_PITCH=16 -DVIS4_SWEEP_ROWS=13 -DVIS4_SWEEP_WAVES=8" \
                  "c4_16x16:-DVIS4_SWEEP_CG=1 -DVIS4_SWEEP_WALK=8 -DVIS4_SWEEP_PITCH=16 -DVIS4_SWEEP_ROWS=15 -DVIS4_SWEEP_WAVES=8" \
                  "c4_32x8:-DVIS4_SWEEP_CG=2 -DVIS4_SWEEP_WALK=8 -DVIS4_SWEEP_PITCH=20 -DVIS4_SWEEP_ROWS=13 -DVIS4_SWEEP_WAVES=8" \
                  "c4_32x12:-DVIS4_SWEEP_CG=2 -DVIS4_SWEEP_WALK=12 -DVIS4_SWEEP_PITCH=20 -DVIS4_SWEEP_ROWS=14 -DVIS4_SWEEP_WAVES=4" \
                  "c4_32x10_p20:-DVIS4_SWEEP_CG=2 -DVIS4_SWEEP_WALK=10 -DVIS4_SWEEP_PITCH=20 -DVIS4_SWEEP_ROWS=13 -DVIS4_SWEEP_WAVES=6" \
                  "c4_64x5:-DVIS4_SWEEP_CG=4 -DVIS4_SWEEP_WALK=10 -DVIS4_SWEEP_PITCH=28 -DVIS4_SWEEP_ROWS=12 -DVIS4_SWEEP_WAVES=6" \
                  "c4_64x4:-DVIS4_SWEEP_CG=4 -DVIS4_SWEEP_WALK=8 -DVIS4_SWEEP_PITCH=28 -DVIS4_SWEEP_ROWS=12 -DVIS4_SWEEP_WAVES=8"
