from .module import BrokenAudio, ShaderAudio
from .spectrogram import ShaderSpectrogram
from .waveform import ShaderWaveform
