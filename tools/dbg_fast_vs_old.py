import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from oracle import binding as O
from tests.helpers import Gpu, gpu_bind_all, oracle_textures, visualizer_inputs
w, h, ssaa = 3840, 2160, 2
u, arrays, params = visualizer_inputs(w, h, seed=51, volume=0.9, bg_size=(1920, 1080)); u.iSSAA = 2.0
gpu = Gpu()
prog, _ = gpu.program("visualizer"); gpu.set_uniforms(prog, u); gpu_bind_all(gpu, prog, arrays, params)
fast = gpu.render_resolve(prog, w, h, ssaa, 2); print(gpu.lib.sfx_last_kernel().decode())
os.environ["SHADERFLOW_VIS_FAST"] = "0"
old = gpu.render_resolve(prog, w, h, ssaa, 2); print(gpu.lib.sfx_last_kernel().decode())
d = np.abs(fast.astype(int) - old.astype(int))
print("fast vs old: hist", np.bincount(d.ravel())[:4], "fraction differing", (d > 0).mean())
for first, last in ((0, 4), (1000, 1004), (2156, 2160)):
    screen = O.render("visualizer", u, oracle_textures(arrays, params), w*ssaa, h*ssaa, rows=(first*ssaa, last*ssaa), threads=8)
    oracle = O.resolve(screen, w, h, 2, rows=(first, last), threads=8)[first:last]
    for name, img in (("fast", fast), ("old", old)):
        dd = np.abs(img[first:last].astype(int) - oracle.astype(int))
        print(first, name, "vs oracle", np.bincount(dd.ravel())[:3], "per channel differing", [(dd[..., c] > 0).mean().round(4) for c in range(3)],
              "signed mean", (img[first:last].astype(int) - oracle.astype(int)).mean().round(4))
rows = np.where((d > 0).any(axis=(1, 2)))[0]
print("rows with differences fast vs old:", len(rows), "of", h, "by row band mean:", [(d[a:a+270] > 0).mean().round(4) for a in range(0, h, 270)])
