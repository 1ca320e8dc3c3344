import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from shaderflow_amd import ShaderScene, synth
from shaderflow_amd.audio import ShaderAudio
from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
from shaderflow_amd.piano import PianoNote
SCROLL = """
void main() {
    vec2 uv = vec2(astuv.x + iSpectrogramOffset, astuv.y);
    vec2 s = sqrt(texture(iSpectrogram, uv).xy)/40.0;
    fragColor = vec4(s, float(iSpectrogramLength)/64.0, 1);
}
"""
pcm, sr = synth.sweep_clip(2.0, 44100), 44100
w, h, fps, frames, length = 96, 54, 60.0, 100, 0.5
class Scroller(ShaderScene):
    def build(self):
        super().build()
        self.audio = ShaderAudio(scene=self, name="iAudio")
        self.audio.load(samples=pcm, samplerate=sr)
        self.spectrogram = ShaderSpectrogram(scene=self, audio=self.audio, length=length, smooth=False)
        self.spectrogram.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
        self.shader.fragment = SCROLL
kw = dict(width=w, height=h, fps=fps, time=frames/fps, ssaa=1, output=bytes)
loop = np.frombuffer(Scroller().main(batch=False, **kw), np.uint8).reshape(frames, h, w, 3)
tape = np.frombuffer(Scroller().main(batch=None, **kw), np.uint8).reshape(frames, h, w, 3)
for k in range(frames):
    d = (loop[k] != tape[k])
    if d.any():
        cols = np.where(d.any(axis=(0, 2)))[0]
        print(k, int(d.sum()), "cols", cols[:6], "...", cols[-3:], "loop", loop[k, 20, cols[0]], "tape", tape[k, 20, cols[0]])
    if k > 40 and d.any(): break

# the tape's own view of the scrolling texture
from shaderflow_amd.tape import FrameTape
from shaderflow_amd import _native as N
scene = Scroller(); scene.initialize()
scene.exporting = scene.freewheel = scene.headless = True; scene.realtime = False
scene.fps, scene.subsample, scene.time = fps, 2, 0.0
from shaderflow_amd.message import ShaderMessage
scene.relay(ShaderMessage.Shader.Compile); scene.resize(width=w, height=h)
for m in scene.modules: m.setup()
scene.set_duration(frames/fps)
tp = FrameTape(scene).prepare(frames); tp.bind_static_uniforms(); N.check(N.lib().sfx_tape_reset(tp.handle))
tp.build(0, 60)
cols = tp.read(N.TAPE_SPECTROGRAM, 60)
scroll = tp.read(N.TAPE_SCROLL, 60)
uni = tp.read(N.TAPE_UNIFORMS, 60)
print("columns max", cols.max(), "scroll max", scroll.max(), "shape", scroll.shape, "offsets", uni[:5, 5])
k = 40
print("frame 40: nonzero columns of scroll", np.where(scroll[k].max(axis=(0, 2)) > 0)[0])
