cd /root/repo
for cfg in "--width 3840 --height 2160 --ssaa 2" "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8" "--width 1920 --height 1080 --ssaa 4" "--width 1280 --height 720 --ssaa 4"; do
  echo "## $cfg"
  bash tools/bench_variants.sh --steps 5 --warmup 2 --no-cpu-baseline --no-export $cfg
done
SHADERFLOW_HIP_LIBRARY=$PWD/build/variants/lib_half4.so python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -x -q -m gpu -k "strip or c4" 2>&1 | tail -2
