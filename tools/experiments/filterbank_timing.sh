#!/bin/bash
# GPU box: the filterbank of the bench command as CSR (bit-exact, what the timed run uses) and on MFMA (k_filterbank_mfma +
# k_filterbank_reduce), kernel-trace averages -> gpurun_out/r05/filterbank.txt (profiles/r05_filterbank.txt)
cd "$(dirname "$0")/../.." || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
for fb in csr mfma; do
  rm -rf gpurun_out/fb_$fb
  SHADERFLOW_FILTERBANK=$fb rocprofv3 --kernel-trace --stats -f csv -d gpurun_out/fb_$fb -o trace -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-export > gpurun_out/fb_$fb.log 2>&1
done
python3 - > gpurun_out/r05/filterbank.txt <<'PY'
import collections, csv, glob
print("filterbank of the bench command (C3: 300 frames x 2 channels x 2049 FFT bins -> 360 bins per launch), rocprofv3 --kernel-trace, bench.py --steps 3 --warmup 1")
print("per call, us, in launch order: the first builds run ALONE (the pipeline primes), the later ones on the audio stream BESIDE the previous batch's render kernel, which fills every CU")
for fb in ("csr", "mfma"):
    calls = collections.defaultdict(list)
    for f in glob.glob(f"gpurun_out/fb_{fb}/**/*kernel_trace.csv", recursive=True):
        for r in sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"])):
            name = r["Kernel_Name"].split("(")[0].replace("sf::", "")
            if "filterbank" in name or "stft" in name:
                calls[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))/1e3)
    for name, us in calls.items():
        print(f"  SHADERFLOW_FILTERBANK={fb:5s} {name:24s} alone (min) {min(us):7.1f}   median {sorted(us)[len(us)//2]:7.1f}   beside the render (max) {max(us):7.1f}   all: " + " ".join(f"{u:.1f}" for u in us))
PY
cat gpurun_out/r05/filterbank.txt
