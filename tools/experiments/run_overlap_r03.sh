# bench lines across configurations (+ the tests that render through tapes)
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_pixels.py tests/test_gpu_scene.py tests/test_gpu_fullsize.py -q -m gpu -x 2>&1 | tail -2
for cfg in "--scene bars" "--scene waveform" "--scene basic" "--width 1920 --height 1080 --ssaa 1" "--width 1920 --height 1080 --ssaa 2" "--width 256 --height 256 --ssaa 1" ""; do
  for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-export $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['launch_ms'])"; done
done
