#!/bin/bash
# bench.py --scene basic under the cameras of basic_tiers.py for the default library and the build/variants/*.so named. GPU box only.
cd "$(dirname "$0")/../.." || exit 1
run() {
  python bench.py --scene basic --steps 20 --warmup 3 --no-cpu-baseline --no-export --camera-zoom $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'zoom', '$2', d['value'], d['roofline']['frac'], d['roofline']['launch_ms'])"
}
for z in ${ZOOM_LIST:-1 0.2}; do run default $z; done
for v in "$@"; do
  export SHADERFLOW_HIP_LIBRARY=build/variants/lib_$v.so
  for z in ${ZOOM_LIST:-1 0.2}; do run $v $z; done
done
