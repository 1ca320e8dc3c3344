#!/bin/bash
# Builds kernel variants (compile-time knobs) into build/variants/*.so; bench them on the GPU box with
#   SHADERFLOW_HIP_LIBRARY=<so> python bench.py …      (tools/bench_variants.sh does that for all of them)
# usage: [UNIT=launch_visualizer_strip] tools/variants.sh "name1:-DFOO=1 -DBAR=2" "name2:…"
# Only the translation unit named by UNIT (default: the strip kernels) is compiled with the flags; the other objects are the default
# build's (shaderflow_amd/csrc/Makefile), so a variant takes one unit's compile time. Variants build side by side.
cd "$(dirname "$0")/../shaderflow_amd/csrc" || exit 1
UNIT=${UNIT:-launch_visualizer_strip}
make >/dev/null 2>&1 || { echo "the default build fails"; exit 1; }
mkdir -p ../../build/variants
FLAGS=$(make -pn 2>/dev/null | sed -n 's/^FLAGS = //p' | head -1 | sed 's/\$(ARCH)/gfx950/')
OTHERS=$(ls ../../build/obj/*.o | grep -v "/$UNIT.o")
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}
  (
    /opt/rocm/bin/hipcc $FLAGS $flags -c $UNIT.hip -o ../../build/variants/$name.$UNIT.o 2>&1 | grep -E "error|spill" ;
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OTHERS ../../build/variants/$name.$UNIT.o -o ../../build/variants/lib_$name.so -lpthread -lhsa-runtime64 &&
    echo "built $name ($flags)"
  ) &
done
wait
