# the strip kernel: its tests + the C3 / C2 / C4 bench lines
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_pixels.py tests/test_gpu_fullsize.py tests/test_gpu_mesa.py tests/test_gpu_scene.py -q -m gpu -x 2>&1 | tail -2
for cfg in "" "--width 1920 --height 1080 --ssaa 1" "--width 1920 --height 1080 --ssaa 2" "--width 7680 --height 4320 --ssaa 4 --frames-per-step 8"; do
  for i in 1 2; do python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-export $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['launch_ms'])"; done
done
