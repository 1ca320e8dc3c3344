cd /root/repo
for so in shaderflow_amd/libshaderflow_hip.so build/variants/lib_dq1.so build/variants/lib_dq4w4.so build/variants/lib_dq8w4.so build/variants/lib_dq4w6.so; do
  for z in 1 0.2; do
    SHADERFLOW_HIP_LIBRARY=$PWD/$so python bench.py --scene basic --camera-zoom $z --frames-per-step 300 --steps 6 --warmup 2 --no-cpu-baseline --no-export --no-live-counters 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$so'.split('/')[-1], 'zoom $z', d['value'], d['roofline']['frac'])"
  done
done
SHADERFLOW_DEFAULT_QUADS=0 python bench.py --scene basic --frames-per-step 300 --steps 6 --warmup 2 --no-cpu-baseline --no-export --no-live-counters 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('quads off zoom 1', d['value'], d['roofline']['frac'])"
