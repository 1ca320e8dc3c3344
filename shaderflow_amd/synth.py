"""
Deterministic synthetic inputs of the benchmark configurations: the reference's
own assets are downloaded from the network (examples/basic/demo.py:19-49) and are not available.
"""
from __future__ import annotations

import math

import numpy as np


def sweep_clip(seconds: float = 60.0, samplerate: int = 44100) -> np.ndarray:
    """(samples, 2) float32: L = 0.5*sin(phase) logarithmic sweep 20 Hz → 20 kHz, R = the same sweep reversed"""
    n = int(round(seconds*samplerate))
    t = np.arange(n, dtype=np.float64)/samplerate
    k = math.log(1000.0)
    phase = 2*math.pi*20.0*seconds/k*(np.exp(t/seconds*k) - 1.0)
    left = 0.5*np.sin(phase)
    return np.stack([left, left[::-1]], axis=1).astype(np.float32)


def background_image(width: int = 1920, height: int = 1080, seed: int = 0) -> np.ndarray:
    """(height, width, 3) uint8: low-frequency value noise over a gradient, top row first (an image file's order)"""
    rng = np.random.default_rng(seed)
    coarse = rng.random((height//40 + 2, width//40 + 2, 3))
    ys = np.linspace(0, coarse.shape[0] - 1.001, height)
    xs = np.linspace(0, coarse.shape[1] - 1.001, width)
    y0, x0 = ys.astype(int), xs.astype(int)
    fy, fx = (ys - y0)[:, None, None], (xs - x0)[None, :, None]
    noise = ((coarse[y0][:, x0]*(1 - fx) + coarse[y0][:, x0 + 1]*fx)*(1 - fy)
             + (coarse[y0 + 1][:, x0]*(1 - fx) + coarse[y0 + 1][:, x0 + 1]*fx)*fy)
    gradient = np.linspace(0.15, 0.85, width)[None, :, None]*np.array([0.9, 0.6, 1.0])[None, None, :]
    fine = rng.random((height, width, 3))*0.08
    return np.clip((0.55*noise + 0.45*gradient + fine)*255.0, 0, 255).astype(np.uint8)
