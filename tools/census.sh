#!/bin/bash
# The strip kernel's ISA census against the CURRENT sources (no GPU needed: hipcc cross-compiles): a -gline-tables-only listing of the
# strip unit → basic blocks (tools/isa_census.py) → instructions per supersample by class and phase (tools/strip_census.py), checked
# against the hardware counters of a bench line when one is given.
#   usage: tools/census.sh [bench line .json] [--per-sample F] [--strip-waves F]   → profiles/$TAG_strip_isa_census.{txt,json}   (TAG=r06)
cd "$(dirname "$0")/.." || exit 1
TAG=${TAG:-r06}
BENCH=$1; shift
( cd shaderflow_amd/csrc && make asm EXTRA=-gline-tables-only >/dev/null 2>&1 && mv launch_visualizer_strip.gfx950.s ../../build/strip.gline.s ) || { echo "listing failed"; exit 1; }
python3 tools/isa_census.py build/strip.gline.s k_visualizer_stripILi72ELi12ELi2ELi9ELi6ELi4ELb0 --dump build/blocks.json > build/isa_census_head.txt || exit 1
{
  echo "# ISA census of k_visualizer_strip<72,12,2,9,6,4,false> — the benchmark's kernel — against the $TAG sources (tools/census.sh $BENCH $*)."
  echo "# listing: make asm EXTRA=-gline-tables-only (launch_visualizer_strip.hip); tools/isa_census.py … --dump blocks.json; tools/strip_census.py blocks.json [bench line]"
  tail -n 2 build/isa_census_head.txt | sed 's/^/# /'
  python3 tools/strip_census.py build/blocks.json $BENCH --json profiles/${TAG}_strip_isa_census.json "$@"
} > profiles/${TAG}_strip_isa_census.txt
tail -n 12 profiles/${TAG}_strip_isa_census.txt
