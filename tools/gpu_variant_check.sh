# parity tests of the visualizer kernels + bench, for every build/variants/lib_*.so (and the default library)
cd /root/repo
for so in shaderflow_amd/libshaderflow_hip.so build/variants/lib_*.so; do
  echo "== $so"
  SHADERFLOW_HIP_LIBRARY=$PWD/$so timeout 900 python -m pytest tests/test_gpu_pixels.py tests/test_gpu_gles.py tests/test_gpu_scene.py -x -q -k "visualizer or full_size or benchmark or end_to_end or fused" 2>&1 | tail -4
done
bash tools/bench_variants.sh --steps 4 --warmup 2 --no-cpu-baseline --no-export
