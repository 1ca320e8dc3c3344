"""Audio modules of the path: PCM source and loudness (module), STFT + filterbank (spectrogram), oscilloscope rows (waveform).
The names a scene imports from the reference's `shaderflow.audio` resolve here as well."""
from shaderflow_amd.audio.module import BrokenAudio, ShaderAudio
from shaderflow_amd.audio.spectrogram import ShaderSpectrogram
from shaderflow_amd.audio.waveform import ShaderWaveform

__all__ = ["BrokenAudio", "ShaderAudio", "ShaderSpectrogram", "ShaderWaveform"]
