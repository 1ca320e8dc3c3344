#!/usr/bin/env python3
"""One 7680x4320 4xSSAA frame through the library named by SHADERFLOW_HIP_LIBRARY → a .npy (argument 1): tools/experiments/c4_tile_sweep.sh
compares every block geometry of the sweep with the shipped one (which tests/test_gpu_fullsize.py holds against the oracle)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from shaderflow_amd import synth                                     # noqa: E402
from tests.helpers import Gpu, gpu_bind_all, smooth_spectrum, visualizer_inputs   # noqa: E402

gpu = Gpu()
w, h, ssaa = 7680, 4320, 4
u, arrays, params = visualizer_inputs(w, h, seed=52, volume=0.8, bg_size=(1920, 1080))
arrays["background"] = np.ascontiguousarray(np.flipud(synth.background_image(1920, 1080)))
arrays["iSpectrogram"] = smooth_spectrum(seed=52)
u.iSSAA = float(ssaa)
prog, _ = gpu.program("visualizer")
gpu.set_uniforms(prog, u)
gpu_bind_all(gpu, prog, arrays, params)
frame = gpu.render_resolve(prog, w, h, ssaa, 2)
print(gpu.lib.sfx_last_kernel().decode())
np.save(sys.argv[1], frame)
