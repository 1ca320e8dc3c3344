import sys, time
sys.path.insert(0, "/root/repo")
import examples.scenes as scenes
scene = scenes.Life()
t=time.perf_counter()
scene.main(width=1920, height=1080, ssaa=1, fps=60.0, time=5.0, output="/dev/null")
print("took", time.perf_counter()-t)
