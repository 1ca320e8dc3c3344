import sys, math
sys.path.insert(0, "/root/repo")
from tests.helpers import Gpu, gpu_bind_all, visualizer_inputs
from tests.test_gpu_pixels import _turn_camera
gpu = Gpu()
w, h, ssaa = 640, 360, 2
u, arrays, params = visualizer_inputs(w, h, seed=31, volume=0.8, bg_size=(384, 216))
_turn_camera(u, roll=200.0, iCameraIsometric=0.3)
u.iSSAA = float(ssaa)
prog, _ = gpu.program("visualizer")
gpu.set_uniforms(prog, u)
gpu_bind_all(gpu, prog, arrays, params)
gpu.ctx.tile_misses()
got = gpu.render_resolve(prog, w, h, ssaa, 2)
gpu.ctx.synchronize()
print("misses", gpu.ctx.tile_misses(), gpu.lib.sfx_last_kernel().decode())
