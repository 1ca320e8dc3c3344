#!/usr/bin/env python3
"""VGPR / SGPR / scratch / LDS / spills of kernels matching a pattern, read from a BUILT library's code object (no recompile).
usage: tools/so_kernel_stats.py lib.so [pattern]"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path

LLVM = Path("/opt/rocm/lib/llvm/bin")
so, pattern = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_visualizer_strip")
with tempfile.TemporaryDirectory() as tmp:
    subprocess.run([LLVM/"llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", so, f"{tmp}/fatbin"], check=True)
    subprocess.run([LLVM/"clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={tmp}/fatbin", f"--output={tmp}/code.o"],
                   check=True, capture_output=True)
    notes = subprocess.run([LLVM/"llvm-readelf", "--notes", f"{tmp}/code.o"], capture_output=True, text=True).stdout
for block in notes.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", block)
    if not name:
        continue
    demangled = subprocess.run(["c++filt", name.group(1)], capture_output=True, text=True).stdout.strip()
    if pattern not in demangled:
        continue
    def get(key):
        m = re.search(rf"\.{key}:\s+(\d+)", block)
        return m.group(1) if m else "?"
    short = re.sub(r"\(.*", "", demangled).replace("sf::", "")
    print(f"{short[:84]:84s} vgpr {get('vgpr_count'):>4s} sgpr {get('sgpr_count'):>4s} scratch {get('private_segment_fixed_size'):>5s} "
          f"lds {get('group_segment_fixed_size'):>6s} spills v{get('vgpr_spill_count')} s{get('sgpr_spill_count')}")
