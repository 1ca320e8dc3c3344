#!/bin/bash
# On the GPU box: bench every build/variants/lib_*.so (plus the default library) and print frames/s + kernel ms
cd "$(dirname "$0")/.." || exit 1
ARGS=${*:---steps 3 --warmup 1 --no-cpu-baseline --no-export}
for so in shaderflow_amd/libshaderflow_hip.so build/variants/lib_*.so; do
  SHADERFLOW_HIP_LIBRARY=$PWD/$so python3 bench.py $ARGS 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read())
print('%-40s %8.1f frames/s   kernel %8.3f ms per %d frames' % ('$so'.split('/')[-1], d['value'], d['roofline']['launch_ms'], d['roofline']['frames_per_launch']))"
done
