#!/usr/bin/env python3
"""On the GPU box: render one case of tests/golden/jit.npz through the translator and save the image to gpurun_out/ (debugging aid).
usage: python tools/dump_translated_case.py <case>"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tests.helpers import Gpu                                   # noqa: E402
from tests.test_gpu_translated import render_case              # noqa: E402

name = sys.argv[1]
out = ROOT/"gpurun_out"
out.mkdir(exist_ok=True)
np.save(out/f"translated_{name}.npy", render_case(Gpu(), name))
print("saved", name)
