"""
The reference's example scenes (examples/basic/demo.py:53-205) as they are written against its API — same class
names, same module construction, same parameters — with two differences forced by the environment:
fragments are named by their registry entry instead of a path into the reference's examples/basic/shaders/, and
assets are synthetic (shaderflow_amd.synth) because the originals are downloads (demo.py:16-49).
With the reference checked out, its own demo.py runs unchanged after `shaderflow_amd.install_alias()`.
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


def make(cls, audio=None, background=None, **fields):
    """Build a scene class with its inputs set before `build()` runs (class attributes, like demo.py's Life)"""
    attrs = {}
    if audio is not None:
        attrs["audio_source"] = audio
    if background is not None:
        attrs["background"] = background
    return type(cls.__name__, (cls,), attrs)(**fields)
