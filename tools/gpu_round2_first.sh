set -x
cd /root/repo
export TMPDIR=/tmp
mkdir -p gpurun_out/r02
timeout 120 build/ubench_valu > gpurun_out/r02/ubench_valu.txt 2>&1
timeout 120 build/ubench_mfma_valu > gpurun_out/r02/ubench_mfma_valu.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r02/pytest_gpu.txt
timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r02/bench_base.json 2> gpurun_out/r02/bench_base.err
SHADERFLOW_HIP_LIBRARY=$PWD/build/variants/lib_timers.so timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/r02/bench_timers.json 2> gpurun_out/r02/bench_timers.err
cat gpurun_out/r02/pytest_gpu.txt; tail -3 gpurun_out/r02/bench_timers.err; cat gpurun_out/r02/bench_base.json
