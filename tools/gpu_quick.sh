# quick GPU check: parity tests that exercise the fused kernels + short benches
cd /root/repo
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_pixels.py tests/test_gpu_gles.py tests/test_gpu_scene.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -15
for scene in visualizer bars; do
  timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-export --scene $scene 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('$scene', d['value'], 'fps; launch', d['roofline']['launch_ms'], 'ms;', d['roofline']['kernel'])"
done
SHADERFLOW_SEPARABLE=0 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-export --scene bars 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); print('bars (PlainShader)', d['value'], 'fps; launch', d['roofline']['launch_ms'], 'ms;', d['roofline']['kernel'])"
