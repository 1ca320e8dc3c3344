import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
from examples.scenes import Waveform, Basic, make
from shaderflow_amd import synth
make(Basic).main(width=1920, height=1080, ssaa=2, fps=60, time=1.0, output="/dev/null")
scene = make(Waveform, audio=(synth.sweep_clip(20.0, 44100), 44100))
pr = cProfile.Profile(); pr.enable()
scene.main(width=1920, height=1080, ssaa=2, fps=60, time=10.0, output="/dev/null")
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
