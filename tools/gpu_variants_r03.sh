# On the GPU box: every build/variants/lib_*.so — one whole-frame C3 parity test, then the bench
cd /root/repo
for so in shaderflow_amd/libshaderflow_hip.so build/variants/lib_*.so; do
  echo "== $so"
  SHADERFLOW_HIP_LIBRARY=$PWD/$so python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -k "c3_whole_frame_single_launch" 2>&1 | tail -1
done
bash tools/bench_variants.sh --steps 6 --warmup 2 --no-cpu-baseline --no-export
