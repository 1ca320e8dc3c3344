# The tests that cover default.glsl's fused kernel + the Basic bench line (round 3's work on k_separable_fused<default>)
set -x
mkdir -p gpurun_out/basic
timeout 1200 python -m pytest tests/test_gpu_fullsize.py::test_basic_whole_frame_4k tests/test_gpu_pixels.py tests/test_gpu_gles.py tests/test_gpu_mesa.py tests/test_gpu_scene.py -x -q -m gpu > gpurun_out/basic/tests.log 2>&1; echo "tests rc=$?"
tail -5 gpurun_out/basic/tests.log
for i in 1 2; do python bench.py --scene basic --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['roofline']['achieved'], d['roofline']['launch_ms'])"; done
