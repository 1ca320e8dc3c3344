"""
The reference's example scenes (examples/basic/demo.py:53-205) as they are written against its API — same class
names, same module construction, same parameters — with two differences forced by the environment:
fragments are named by their registry entry instead of a path into the reference's examples/basic/shaders/, and
assets are synthetic (shaderflow_amd.synth) because the originals are downloads (demo.py:16-49).
With the reference checked out, its own demo.py runs unchanged after `shaderflow_amd.install_alias()`.
`Plasma` is not from the reference: it shows a user-written fragment going through the run-time translator.
"""
from __future__ import annotations

import math
from pathlib import Path
from typing import Optional

import numpy as np

from shaderflow_amd import synth
from shaderflow_amd.dynamics import ShaderDynamics
from shaderflow_amd.scene import ShaderScene
from shaderflow_amd.shader import ShaderProgram
from shaderflow_amd.texture import ShaderTexture


class Basic(ShaderScene):
    """Simplest ShaderScene (demo.py:53-55): the built-in default fragment"""
    ...


class ShaderToy(ShaderScene):
    def build(self):
        self.shader.fragment = "shadertoy"


class MultiShader(ShaderScene):
    """Two shaders acting together (demo.py:67-89)"""
    def build(self):
        self.child = ShaderProgram(scene=self, name="child")
        self.child.fragment = "multi_child"
        self.shader.fragment = "multi_main"


class Dynamics(ShaderScene):
    """Second order system driven from python every frame (demo.py:114-129): frame-loop only"""
    background: Optional[np.ndarray] = None

    def build(self):
        image = self.background if self.background is not None else synth.background_image(480, 270)
        ShaderTexture(scene=self, name="background").from_numpy(image)
        self.dynamics = ShaderDynamics(scene=self, name="iShaderDynamics", frequency=4)
        self.shader.fragment = "dynamics"

    def update(self):
        self.dynamics.target = 0.5*(1 + np.sign(np.sin(2*math.pi*self.time*0.5)))


class _AudioScene(ShaderScene):
    """Scenes below take `audio=` as a WAV path or a (samples (n, channels) float32, samplerate) pair"""
    audio_source = None

    def _load_audio(self):
        from shaderflow_amd.audio import ShaderAudio
        self.audio = ShaderAudio(scene=self, name="iAudio")
        source = self.audio_source
        if isinstance(source, (str, Path)):
            self.audio.file = source
        elif source is not None:
            samples, samplerate = source
            self.audio.load(samples=samples, samplerate=samplerate)


class Waveform(_AudioScene):
    """Audio waveform oscilloscope (demo.py:157-166)"""
    def build(self):
        from shaderflow_amd.audio.waveform import ShaderWaveform
        self._load_audio()
        self.waveform = ShaderWaveform(scene=self, audio=self.audio, smooth=False)
        self.shader.fragment = "waveform"


class MusicBars(_AudioScene):
    """Basic music bars (demo.py:170-184)"""
    def build(self):
        from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
        from shaderflow_amd.piano import PianoNote
        self._load_audio()
        self.spectrogram = ShaderSpectrogram(scene=self, audio=self.audio, length=0)
        self.spectrogram.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(18000), piano=True)
        self.shader.fragment = "bars"


class Visualizer(_AudioScene):
    """Radial bars music visualizer (demo.py:188-205) — the benchmark scene"""
    background: Optional[np.ndarray] = None

    def build(self):
        from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
        from shaderflow_amd.audio.waveform import ShaderWaveform
        from shaderflow_amd.piano import PianoNote
        self._load_audio()
        self.waveform = ShaderWaveform(scene=self, audio=self.audio)
        self.spectrogram = ShaderSpectrogram(scene=self, length=0, audio=self.audio, smooth=False)
        self.spectrogram.from_notes(start=PianoNote.from_frequency(20), end=PianoNote.from_frequency(14000), piano=True)
        image = self.background if self.background is not None else synth.background_image(1920, 1080)
        self.back = ShaderTexture(scene=self, name="background").from_numpy(image)
        self.shader.fragment = "visualizer"


class Multipass(ShaderScene):
    """Multi layers done on a single shader (demo.py:93-99)"""
    background: Optional[np.ndarray] = None

    def build(self):
        image = self.background if self.background is not None else synth.background_image(480, 270)
        ShaderTexture(scene=self, name="background").from_numpy(image)
        self.shader.texture.layers = 2
        self.shader.fragment = "multipass"


class MotionBlur(ShaderScene):
    """Poor's man Motion Blur (demo.py:103-110)"""
    background: Optional[np.ndarray] = None

    def build(self):
        image = self.background if self.background is not None else synth.background_image(480, 270)
        ShaderTexture(scene=self, name="background").from_numpy(image)
        self.shader.texture.temporal = 10
        self.shader.texture.layers = 2
        self.shader.fragment = "motionblur"


class RayMarch(ShaderScene):
    """Ray Marching demo (demo.py:215-219)"""
    def build(self):
        self.shader.fragment = "raymarch"


class Life(ShaderScene):
    """Conway's Game of Life on the GPU (demo.py:223-247): a float simulation texture with ten frames of history"""
    life_period: int = 6
    shard_warmup = None                  # every generation depends on all earlier ones: a shard renders from frame 0

    def setup(self):
        width, height = 192, 108
        random = np.random.randint(0, 2, (width, height), dtype=bool)
        self.simulation.texture.size = (width, height)
        self.simulation.texture.write(random.astype(np.float32), temporal=1)

    def build(self):
        self.simulation = ShaderProgram(scene=self, name="iLife")
        self.simulation.texture.temporal = 10
        self.simulation.texture.filter = "nearest"
        self.simulation.texture.dtype = "f4"
        self.simulation.texture.components = 1
        self.simulation.texture.track = False
        self.simulation.fragment = "life_simulation"
        self.shader.fragment = "life_visuals"

    def pipeline(self):
        from shaderflow_amd.variable import Uniform
        yield from ShaderScene.pipeline(self)
        yield Uniform("int", "iLifePeriod", self.life_period)


class Mandelbrot(ShaderScene):
    """Mandelbrot fractal (examples/fractals/fractals.py:7-10)"""
    def build(self):
        self.shader.fragment = "mandelbrot"


class Tetration(ShaderScene):
    """Complex tetration fractal (examples/fractals/fractals.py:12-15)"""
    def build(self):
        self.shader.fragment = "tetration"


class Video(ShaderScene):
    """Video as a texture (shaderflow/video.py + examples/basic/shaders/video.frag; demo.py has no scene for it):
    `clip` = (frames (n, h, w, 3) uint8, fps) or a path"""
    clip = None

    def build(self):
        from shaderflow_amd.video import ShaderVideo
        if isinstance(self.clip, (str, Path)):
            self.video = ShaderVideo(scene=self, path=self.clip)
        else:
            frames, fps = self.clip if self.clip is not None else (synth.background_image(64, 36)[None].repeat(4, 0), 30.0)
            self.video = ShaderVideo(scene=self, frames=frames, fps=fps)
        self.shader.fragment = "video"


class Plasma(ShaderScene):
    """A scene with a fragment of its own (not one of the reference's): the GLSL below is translated to HIP C++, compiled
    with hipcc on first use and cached (shaderflow_amd/glsl2hip.py). Scene-defined uniforms come from `pipeline()` like in the
    reference; being a stock scene otherwise, it batches through the clock tape."""
    FRAGMENT = """
        // sum of rotating plane waves, coloured through the prelude's hsv2rgb
        #define WAVES 5
        float wave(vec2 p, float k) {
            vec2 direction = vec2(cos(k*1.3), sin(k*1.3));
            return sin(dot(direction, p)*(3.0 + k) + iTime*(0.6 + 0.2*k));
        }
        void main() {
            GetCamera(iCamera);
            vec2 p = iCamera.gluv*rotate2d(0.1*iTime);
            float sum = 0;
            for (int k = 0; k < WAVES; k++)
                sum += wave(p, float(k))/WAVES;
            vec3 colour = hsv2rgb(vec3(TAU*fract(0.5*sum + 0.05*iTime), 0.75, 0.6 + 0.4*sum));
            colour *= 1 - 0.3*smoothstep(0.5, 1.5, length(iCamera.gluv));
            fragColor = vec4(colour, 1);
        }
    """

    def build(self):
        super().build()
        self.shader.fragment = self.FRAGMENT


class Bloom(ShaderScene):
    """A picture with a glow: a fragment of its own that gathers the bright parts around every pixel with a loop of taps — the kind
    of fragment whose sampler the kernels serve from an LDS tile (DESIGN.md §9 "Translated fragments"; `SHADERFLOW_JIT_TILE=0`
    renders the same frames without it)"""
    background: Optional[np.ndarray] = None
    FRAGMENT = """
        uniform float iGlow = 0.6;
        void main() {
            vec2 texel = 1.0/vec2(textureSize(background, 0));
            float reach = 2.0 + 1.5*sin(iTime);
            vec3 glow = vec3(0);
            float total = 0;
            for (int x = -4; x <= 4; x++) {
                for (int y = -4; y <= 4; y++) {
                    float weight = exp(-float(x*x + y*y)/8.0);
                    vec3 tap = texture(background, astuv + vec2(x, y)*texel*reach).rgb;
                    glow += weight*max(tap - 0.5, 0.0);
                    total += weight;
                }
            }
            fragColor = vec4(texture(background, astuv).rgb + iGlow*2.0*glow/total, 1);
        }
    """

    def build(self):
        super().build()
        image = self.background if self.background is not None else synth.background_image(640, 360)
        self.back = ShaderTexture(scene=self, name="background").from_numpy(image)
        self.shader.fragment = self.FRAGMENT


def make(cls, audio=None, background=None, **fields):
    """Build a scene class with its inputs set before `build()` runs (class attributes, like demo.py's Life)"""
    attrs = {}
    if audio is not None:
        attrs["audio_source"] = audio
    if background is not None:
        attrs["background"] = background
    return type(cls.__name__, (cls,), attrs)(**fields)
