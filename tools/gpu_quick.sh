# quick GPU check of the visualizer paths: parity tests that exercise the fused kernels + a short bench (A/B with the old kernel)
cd /root/repo
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_pixels.py tests/test_gpu_gles.py tests/test_gpu_scene.py -x -q 2>&1 | tail -15 > gpurun_out/r02/pytest_quick.txt
cat gpurun_out/r02/pytest_quick.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>gpurun_out/r02/bench_fast.err | tail -1 > gpurun_out/r02/bench_fast.json
SHADERFLOW_VIS_FAST=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r02/bench_old.json
python3 -c "
import json
for n in ('fast','old'):
    d=json.load(open('gpurun_out/r02/bench_%s.json'%n)); print(n, d['value'], d['roofline']['launch_ms'])
"
tail -3 gpurun_out/r02/bench_fast.err
