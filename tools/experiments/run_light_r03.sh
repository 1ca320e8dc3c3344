# bars / waveform bench lines + their kernel tests
for s in bars waveform; do for i in 1 2; do python bench.py --scene $s --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$s', d['value'], d['roofline']['achieved'], d['roofline']['launch_ms'])"; done; done
timeout 900 python -m pytest tests/test_gpu_pixels.py -q -m gpu -k "separable or run" 2>&1 | tail -2
