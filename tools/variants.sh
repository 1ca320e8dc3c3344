#!/bin/bash
# Builds kernel variants (compile-time knobs) into build/variants/*.so; bench them on the GPU box with
#   SHADERFLOW_HIP_LIBRARY=<so> python bench.py …
# usage: tools/variants.sh "name1:-DFOO=1 -DBAR=2" "name2:…"
cd "$(dirname "$0")/../shaderflow_amd/csrc" || exit 1
mkdir -p ../../build/variants
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  make -B EXTRA="$flags" OUT=../../build/variants/lib_$name.so 2>&1 | grep -E "error" 
  echo "built $name ($flags)"
done
make -B >/dev/null 2>&1   # restore the default library
