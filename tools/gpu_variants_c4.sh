# parity of the default build, then the variants at C3 and C4
cd /root/repo
mkdir -p gpurun_out/r02
python -m pytest tests/test_gpu_pixels.py tests/test_gpu_gles.py -m gpu -x -q 2>&1 | tail -5
{ echo "## C3"; bash tools/bench_variants.sh --steps 4 --warmup 2 --no-cpu-baseline --no-export
  echo "## C4"; bash tools/bench_variants.sh --steps 3 --warmup 1 --no-cpu-baseline --no-export --width 7680 --height 4320 --ssaa 4 --frames-per-step 8; } > gpurun_out/r02/variants_strip.txt 2>&1
cat gpurun_out/r02/variants_strip.txt
