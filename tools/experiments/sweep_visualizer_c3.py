#!/usr/bin/env python3
"""
A one-off sweep of the benchmark's kernel against the oracle on WHOLE 3840x2160 2xSSAA frames: loudness, flash, time (zoom, offsets, blur
radius), the spectrogram (silence, a smooth column, white noise), cameras — the kind of sweep that found hsv2rgb's singular line in the
Basic scene (round 6). Prints one line per case: the kernel, the waves on the per-sample path, the histogram of |HIP - oracle|.
usage (GPU box): python tools/experiments/sweep_visualizer_c3.py [cases …]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import binding as O                                     # noqa: E402
from shaderflow_amd import synth                                     # noqa: E402
from tests.helpers import Gpu, gpu_bind_all, oracle_textures, smooth_spectrum, visualizer_inputs   # noqa: E402

CASES = [
    dict(volume=0.0, std=0.0, time=0.0, spectrum="zero"), dict(volume=0.2, std=0.05, time=0.31, spectrum="smooth"),
    dict(volume=0.5, std=0.35, time=7.77, spectrum="smooth"), dict(volume=0.97, std=1.0, time=33.3, spectrum="smooth"),
    dict(volume=1.3, std=0.6, time=59.9, spectrum="noise"), dict(volume=0.35, std=0.35, time=12.5, spectrum="peak"),
    dict(volume=0.6, std=0.2, time=3.0, spectrum="smooth", zoom=0.8), dict(volume=0.6, std=0.2, time=3.0, spectrum="smooth", zoom=1.25, pan=(0.07, -0.04)),
]


def main() -> None:
    gpu = Gpu()
    w, h, ssaa = 3840, 2160, 2
    background = np.ascontiguousarray(np.flipud(synth.background_image(1920, 1080)))
    worst = 0
    for n, case in enumerate(CASES):
        u, arrays, params = visualizer_inputs(w, h, seed=100 + n, volume=case["volume"], std=case["std"], time=case["time"], bg_size=(1920, 1080))
        arrays["background"] = background
        if case["spectrum"] == "zero":
            arrays["iSpectrogram"] = np.zeros((115, 1, 2), np.float32)
        elif case["spectrum"] == "smooth":
            arrays["iSpectrogram"] = smooth_spectrum(seed=100 + n)
        elif case["spectrum"] == "peak":
            column = np.full((115, 1, 2), 1e-4, np.float32); column[40:43, 0, 0] = (900.0, 2500.0, 700.0); column[77:79, 0, 1] = (1500.0, 1200.0)
            arrays["iSpectrogram"] = column
        u.iSSAA = float(ssaa)
        if "zoom" in case:
            u.iCameraZoom = case["zoom"]
        if "pan" in case:
            u.iCameraPosition[0], u.iCameraPosition[1] = case["pan"]
        prog, _ = gpu.program("visualizer")
        gpu.set_uniforms(prog, u)
        gpu_bind_all(gpu, prog, arrays, params)
        gpu.ctx.tile_misses()
        got = gpu.render_resolve(prog, w, h, ssaa, 2)
        per_sample = gpu.ctx.tile_misses()
        textures = oracle_textures(arrays, params)
        histogram = np.zeros(4, np.int64)
        for first in range(0, h, 216):
            screen = O.render("visualizer", u, textures, w*ssaa, h*ssaa, rows=(first*ssaa, (first + 216)*ssaa), threads=16)
            want = O.resolve(screen, w, h, 2, rows=(first, first + 216), threads=16)[first:first + 216]
            d = np.abs(got[first:first + 216].astype(int) - want.astype(int))
            histogram += np.bincount(np.minimum(d.ravel(), 3), minlength=4)
        worst = max(worst, int(np.nonzero(histogram)[0].max()))
        print(f"{case}: {gpu.lib.sfx_last_kernel().decode()}, {per_sample} of 57600 waves per sample, |d| = 0/1/2/3+: {histogram.tolist()}", flush=True)
    print("worst |d|:", worst)


if __name__ == "__main__":
    main()
