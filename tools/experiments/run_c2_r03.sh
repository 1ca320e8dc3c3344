cd /root/repo; export TMPDIR=/tmp
for i in 1 2; do python bench.py --width 1920 --height 1080 --ssaa 1 --steps 10 --warmup 2 --no-cpu-baseline --no-export 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'])"; done
rm -rf gpurun_out/c2t; rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/c2t -o t -- python3 bench.py --width 1920 --height 1080 --ssaa 1 --steps 3 --warmup 1 --no-cpu-baseline --no-export > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/c2t/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]:
        print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)
PY
timeout 600 python -m pytest tests/test_gpu_pixels.py -q -m gpu -k "tent or resolve" 2>&1 | tail -2
