#!/bin/bash
# tools/variants.sh with the builds in parallel (one hipcc each); leaves the default library alone.
# usage: tools/variants_par.sh "name1:-DFOO=1 -DBAR=2" "name2:…"
cd "$(dirname "$0")/../shaderflow_amd/csrc" || exit 1
mkdir -p ../../build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  ( make -B EXTRA="$flags" OUT=../../build/variants/lib_$name.so 2>&1 | grep -E "error"; echo "built $name ($flags)" ) &
done
wait
