#!/usr/bin/env python3
"""How the oracle's row bands scale with threads on the GPU box's host (bench.py cpu_baseline reported 9x of one thread on 256
threads in round 3): cgroup CPU quota, then seconds per 512-row band of the C3 visualizer frame at 1 … 256 threads."""
import os
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import numpy as np  # noqa: E402

from oracle import binding as O  # noqa: E402
from tests.helpers import oracle_textures, visualizer_inputs  # noqa: E402

for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
        print(path, "=", open(path).read().strip())
    except OSError:
        pass
print("os.cpu_count()", os.cpu_count(), "sched_getaffinity", len(os.sched_getaffinity(0)), "loadavg", os.getloadavg())
w, h, s = 3840, 2160, 2
u, arrays, params = visualizer_inputs(w, h, seed=3, volume=0.8, bg_size=(1920, 1080))
u.iSSAA = 2.0
textures = oracle_textures(arrays, params)
rows = 512
base = None
for threads in (1, 8, 16, 32, 64, 128, 256):
    band = rows if threads > 1 else 16
    t = time.perf_counter()
    O.render("visualizer", u, textures, w*s, h*s, rows=(1000, 1000 + band), threads=threads)
    took = (time.perf_counter() - t)*rows/band
    base = base or took
    print(f"{threads:4d} threads: {took:7.2f} s per {rows} supersample rows  ({base/took:6.1f}x one thread)", flush=True)
