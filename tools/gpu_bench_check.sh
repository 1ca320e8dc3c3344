cd /root/repo
mkdir -p gpurun_out/r02
timeout 600 python bench.py --steps 4 --warmup 2 --cpu-seconds 8 > gpurun_out/r02/bench_new.json 2> gpurun_out/r02/bench_new.err; tail -3 gpurun_out/r02/bench_new.err; cat gpurun_out/r02/bench_new.json
timeout 900 python -m pytest tests/test_gpu_distributed.py -q -x -k bench 2>&1 | tail -15
