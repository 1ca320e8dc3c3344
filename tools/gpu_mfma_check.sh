cd /root/repo; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_audio.py -q -x 2>&1 | tail -3
rm -rf gpurun_out/mfma && mkdir -p gpurun_out/mfma
SHADERFLOW_FILTERBANK=mfma rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/mfma -o t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export > gpurun_out/mfma/log.txt 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/mfma/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "filterbank" in r["Name"]: print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, "us")
PY
