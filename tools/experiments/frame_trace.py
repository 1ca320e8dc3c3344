import sys, time
sys.path.insert(0, "/root/repo")
import examples.scenes as scenes
from shaderflow_amd import synth, _native as N
from shaderflow_amd.exporting import ExportingHelper
import ctypes as C
marks = []
orig_pipe = ExportingHelper.pipe
def pipe(self, turbo=False):
    t0 = time.perf_counter()
    slot = self.frame % self.slots
    N.check(N.lib().sfx_ring_read_async(self.ring, self.scene._final.texture.texture.handle, slot))
    t1 = time.perf_counter()
    N.check(N.lib().sfx_ring_pipe(self.ring, slot, self.fileno))
    t2 = time.perf_counter()
    marks.append((t1 - t0, t2 - t1))
ExportingHelper.pipe = pipe
scene = scenes.make(scenes.Dynamics, background=synth.background_image(1920, 1080, seed=0))
scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=2.0, output="/dev/null", batch=False)
import numpy as np
m = np.array(marks)*1e6
print("read_async us: median %.0f mean %.0f; pipe us: median %.0f" % (np.median(m[:,0]), m[:,0].mean(), np.median(m[:,1])))
print(m[40:52].round())
