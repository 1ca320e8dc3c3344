# the two-bank tape / two-stream two-pass path: the bench lines they move (+ the tests that render through tapes)
cd /root/repo
timeout 1500 python -m pytest tests/test_gpu_audio.py tests/test_gpu_scene.py tests/test_gpu_distributed.py tests/test_gpu_fullsize.py tests/test_gpu_translated.py -q -m gpu -x 2>&1 | tail -2
for cfg in "--width 1920 --height 1080 --ssaa 1" "--width 1280 --height 720 --ssaa 1" "--width 3840 --height 2160 --ssaa 1 --frames-per-step 20" "--scene bars --ssaa 1" "--scene basic --ssaa 1" ""; do
  for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-export $cfg 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$cfg', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['launch_ms'])"; done
done
